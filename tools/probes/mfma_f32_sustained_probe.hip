// mfma_f32_sustained_probe -- what exact-f32 matrix rate does the chip SUSTAIN?  (hipcc --offload-arch=gfx950 -O3)
// Nominal: 256 CUs x 4 SIMDs x one v_mfma_f32_32x32x2_f32 (4096 FLOP) per 64 cycles at 2.4 GHz = 157.3 TFLOP/s.  Like the
// 16-bit probe (mfma_sustained_probe.hip): nothing but register-resident MFMAs, 4 independent accumulators per wave
// (the 2 x 2 block layout of gemm_f32_mfma_kernel), 1 or 2 waves per SIMD, random operands in [-1, 1) or zeros,
// 64 .. 256 CUs -- the power-limited ceiling the ECAPA-TDNN products (BASELINE configs[4]) can be held against.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void mfma_f32_loop(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters,
                                                     int zero) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  uint32_t s = seed[tid % 4096] * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = zero ? 0.f : (float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f;
    s = s * 1664525u + 1013904223u;
    b[i] = zero ? 0.f : (float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f;
  }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[2 * r + j], a[2 * r + i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) t += acc[i][j][e];
  out[tid] = t;
}

int main() {
  uint32_t* seed;
  float* out;
  hipMalloc(&seed, 4096 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  uint32_t h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (uint32_t)rand();
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("# register-resident v_mfma_f32_32x32x2_f32 only; TFLOP/s over the whole chip (nominal peak 157.3 at 2.4 GHz)\n");
  printf("%6s %10s %8s %10s %10s\n", "CUs", "waves/SIMD", "data", "ms", "TFLOP/s");
  for (int zero = 0; zero < 2; ++zero)
    for (int wps = 1; wps <= 2; ++wps)
      for (int cus = 64; cus <= 256; cus += 64) {
        const int iters = 24000 / wps;               // 16 MFMAs per iteration and wave
        hipLaunchKernelGGL(mfma_f32_loop, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters / 8, zero);   // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_f32_loop, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters, zero);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 4 * wps * (double)iters * 16 * 4096.0;
        printf("%6d %10d %8s %10.3f %10.1f\n", cus, wps, zero ? "zeros" : "random", ms, flops / (ms * 1e-3) / 1e12);
      }
  return 0;
}
