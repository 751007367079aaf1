// Issue-to-landing latency of one global_load_lds_dwordx4 wave-instruction (1 KiB) on gfx950:
// L1- / L2-resident source (small buffers re-read) vs HBM-resident source (1 GiB, every access new lines).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
__global__ __launch_bounds__(64) void probe(const char* src, size_t stride, size_t span, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  const char* p = src + lane * 16;
  size_t off = (size_t)blockIdx.x * 65536 % span;
  unsigned long long total = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_global_load_lds((gvoid_t*)(p + off), (lvoid_t*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    total += __builtin_readcyclecounter() - t0;
    off = (off + stride) % span;
  }
  if (lane == 0) out[blockIdx.x] = total;
}
int main() {
  const size_t big = 1ull << 30;
  char* d; unsigned long long* o; unsigned long long h[256];
  hipMalloc(&d, big + (1 << 20)); hipMemset(d, 1, big + (1 << 20)); hipMalloc(&o, 256 * 8);
  const int iters = 2000;
  struct { const char* name; size_t stride, span; int grid; } cases[] = {
      {"L1-resident, 1 wave on the chip", 1024, 8 << 10, 1},
      {"L2-resident, 1 wave on the chip", 64 << 10, 2 << 20, 1},
      {"HBM stream,  1 wave on the chip", (1 << 20) + 4096, big, 1},
      {"L2-resident, 1 wave per CU (256)", 64 << 10, 2 << 20, 256},
      {"HBM stream,  1 wave per CU (256)", (1 << 20) + 4096, big, 256}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(c.grid), dim3(64), 4096, 0, d, c.stride, c.span, iters, o);
    hipLaunchKernelGGL(probe, dim3(c.grid), dim3(64), 4096, 0, d, c.stride, c.span, iters, o);
    hipMemcpy(h, o, c.grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < c.grid; ++i) s += (double)h[i] / iters;
    printf("%-36s %7.0f cycles = %6.0f ns per 1 KiB LDS-DMA load\n", c.name, s / c.grid, s / c.grid / 2.4);
  }
  return 0;
}
