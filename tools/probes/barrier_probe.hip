// Cost of one s_barrier round for an 8-wave workgroup (and of __syncthreads) on gfx950, in shader clocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* cyc, unsigned long long* rt) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) __builtin_amdgcn_s_barrier();
    else if (MODE == 1) __syncthreads();
    else asm volatile("s_nop 0");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}
int main() {
  unsigned long long *c, *r, hc, hr;
  hipMalloc(&c, 8 * 256); hipMalloc(&r, 8 * 256);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, device clock %d kHz\n", rate, clk);
  const int iters = 100000;
  const char* names[3] = {"s_barrier", "__syncthreads", "s_nop loop"};
  for (int m = 0; m < 3; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(512), 0, 0, iters, c, r);
      if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(512), 0, 0, iters, c, r);
      if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(512), 0, 0, iters, c, r);
    }
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost); hipMemcpy(&hr, r, 8, hipMemcpyDeviceToHost);
    printf("%-14s: %.1f counter ticks / iteration, %.1f ns / iteration\n", names[m], (double)hc / iters, (double)hr / iters * 1e6 / rate);
  }
  return 0;
}
