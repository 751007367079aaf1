// Does a kernel that uses scratch (private segment) cost more per dispatch?  Two kernels with the same body, one keeps
// a small array in scratch (volatile indexing); N back-to-back launches of each, timed with events.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_plain(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
__global__ void k_scratch(float* p, int n) {
  volatile float a[24];
  for (int i = 0; i < 24; ++i) a[i] = p[(threadIdx.x + i) & 255];
  float v = 0.f;
  for (int i = 0; i < n; ++i) v = v * 1.0001f + a[(i + threadIdx.x) % 24];
  p[threadIdx.x] = v;
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {1, 256, 1024}) {
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 2000; ++i) {
          if (which == 0) hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, d, 8);
          else hipLaunchKernelGGL(k_scratch, dim3(grid), dim3(256), 0, 0, d, 8);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("grid %5d %s: %.2f us per launch\n", grid, which ? "scratch" : "plain  ", ms * 1e3 / 2000);
      }
    }
  }
  return 0;
}
