// Probe ds_read_b64_tr_b16 semantics: each lane supplies an 8-byte-aligned LDS address; within a
// 16-lane group the 16x4 elements are transposed.  Prints what lane l receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint16_t* in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // group g reads the 4x16 block: rows k = g*8 + {0..3}, cols 32 + (0..15); lane i -> row i/4, cols (i%4)*4
  const uint32_t addr = (uint32_t)(uintptr_t)(&lds[(g * 8 + i / 4) * 128 + 32 + (i % 4) * 4]);
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t h[64 * 128], o[256];
  for (int k = 0; k < 64; ++k) for (int m = 0; m < 128; ++m) h[k * 128 + m] = (uint16_t)(k * 256 + m);  // hi byte k, lo byte m
  uint16_t *din, *dout;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int i = l & 15, g = l >> 4;
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" (k=%d,m=%d)", o[l * 4 + j] >> 8, o[l * 4 + j] & 255);
      if ((o[l * 4 + j] >> 8) != g * 8 + j || (o[l * 4 + j] & 255) != 32 + i) bad++;
    }
    printf("\n");
  }
  printf("expected lane(i,g) elem j = (k=g*8+j, m=32+i): mismatches=%d\n", bad);
  return 0;
}
