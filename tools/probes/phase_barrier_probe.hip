// phase_barrier_probe -- how much of the matrix pipe does the ANTI-PHASE barrier structure of the phased GEMM kernels cost
// by itself?  Eight waves per workgroup (one workgroup per CU), waves w and w + 4 share a SIMD and form two groups; every
// phase is  { "read segment": nothing here }  s_barrier  { PH MFMAs, s_setprio 1 }  s_barrier ; group 1 runs one barrier
// behind group 0, so on every SIMD one wave multiplies while the other sits in its (empty) read segment.  No LDS, no
// memory: the MFMA-only variant of gemm16_phased_256x256_kernel (tools/gemm_attrib.py) measured 33.4 us against 26.5 us of
// sustained matrix time for FFN1 -- this probe sweeps PH = MFMAs per phase (the kernel uses 16) and two ways of shortening
// the idle gap at a barrier: splitting the phase's MFMAs around the barrier, and dropping the second barrier of a phase.
//   hipcc --offload-arch=gfx950 -O3 -o phase_barrier_probe phase_barrier_probe.hip && ./phase_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: barrier, PH MFMAs, barrier (the kernel's structure)      MODE 1: one barrier per phase (barrier, PH MFMAs)
// MODE 2: barrier, PH/2 MFMAs, barrier, PH/2 MFMAs (a barrier in the middle of the MFMA block)
// MODE 3: no barriers at all (two waves per SIMD free-running: the sustained rate of this instruction mix)
template <int PH, int MODE>
__global__ __launch_bounds__(512) void phase_kernel(const uint32_t* __restrict__ seed, float* __restrict__ out, int phases) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (threadIdx.x >> 6) >> 2;
  f16x8 a[4], b[4];
  uint32_t s = seed[tid % 4096] * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      a[i][e] = (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
      s = s * 1664525u + 1013904223u;
      b[i][e] = (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
    }
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE != 3 && grp == 1) __builtin_amdgcn_s_barrier();
#pragma unroll 1
  for (int p = 0; p < phases; ++p) {
    if (MODE != 3) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < PH; ++m) {
      if (MODE == 2 && m == PH / 2) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[(m >> 2) & 7][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 3], b[(m >> 2) & 3], acc[(m >> 2) & 7][m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) __builtin_amdgcn_s_barrier();
  }
  if (MODE != 3 && grp == 0) __builtin_amdgcn_s_barrier();
  float t = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = t;
}

template <int PH, int MODE>
static void run(const uint32_t* seed, float* out, const char* what) {
  const int total_mfma = 1 << 18;                       // per wave
  const int phases = total_mfma / PH;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((phase_kernel<PH, MODE>), dim3(256), dim3(512), 0, 0, seed, out, phases);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double flops = 256.0 * 8 * total_mfma * 16384.0;
  printf("%3d MFMAs per phase, %-46s %8.2f ms  %7.1f TFLOP/s\n", PH, what, ts[ts.size() / 2], flops / (ts[ts.size() / 2] * 1e-3) / 1e12);
}

int main() {
  uint32_t* seed; float* out;
  hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
  std::vector<uint32_t> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 1234567u * (i + 1);
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<16, 3>(seed, out, "no barriers (two free-running waves per SIMD)");
  run<16, 0>(seed, out, "barrier | MFMAs | barrier (the kernel)");
  run<32, 0>(seed, out, "barrier | MFMAs | barrier");
  run<64, 0>(seed, out, "barrier | MFMAs | barrier");
  run<16, 1>(seed, out, "one barrier per phase");
  run<32, 1>(seed, out, "one barrier per phase");
  run<16, 2>(seed, out, "barrier | half | barrier | half");
  run<32, 2>(seed, out, "barrier | half | barrier | half");
  return 0;
}
