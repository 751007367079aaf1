// Per-CU throughput of the two HBM/L2 -> CU paths on gfx950, from an L2-resident buffer:
//   (a) global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction)   (b) global_load_dwordx4 -> VGPR
// 8 waves per workgroup, one workgroup per CU, each wave streams its own 1 KiB-strided slices.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* src, size_t span, int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* p = src + ((size_t)blockIdx.x * 8 + wave) * 65536 % span + lane * 16;
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* q = p + (size_t)(it & 7) * 8192;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(q + j * 1024), (lvoid_t*)(lds + wave * 16384 + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 v[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) v[j] = *reinterpret_cast<const uint4*>(q + j * 1024);
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) acc += __uint_as_float(v[j].x ^ v[j].w);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.f) sink[0] = acc;
}
template <int MODE, int DEPTH>
static void run(const char* name, const char* d, size_t span, int grid, unsigned long long* c, float* sink) {
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(grid), dim3(512), 131072, 0, d, span, iters, c, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(grid), dim3(512), 131072, 0, d, span, iters, c, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * 8 * iters * DEPTH * 1024.0;
  printf("%-28s depth %2d grid %3d: %7.1f us  %6.2f TB/s total  %6.1f GB/s per CU\n", name, DEPTH, grid, ms * 1e3, bytes / ms / 1e9,
         bytes / grid / ms / 1e6);
}
int main() {
  const size_t span = 64u << 20;
  char* d; unsigned long long* c; float* sink;
  hipMalloc(&d, span + (1 << 20)); hipMemset(d, 1, span + (1 << 20)); hipMalloc(&c, 4096 * 8); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int grid : {1, 32, 256}) {
    run<0, 6>("global_load_lds_dwordx4", d, span, grid, c, sink);
    run<0, 12>("global_load_lds_dwordx4", d, span, grid, c, sink);
    run<1, 6>("global_load_dwordx4 -> VGPR", d, span, grid, c, sink);
    run<1, 12>("global_load_dwordx4 -> VGPR", d, span, grid, c, sink);
  }
  return 0;
}
