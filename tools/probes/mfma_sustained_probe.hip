// mfma_sustained_probe -- what dense 16-bit MFMA rate does the chip SUSTAIN?  (build: hipcc --offload-arch=gfx950 -O3)
//
// The 2.5 PFLOP/s figure is 256 CUs x 4 SIMDs x one v_mfma_f32_16x16x32 per 16 cycles at 2.4 GHz.  Under matrix load the
// chip clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"), so the rate a GEMM can be held against is
// lower and depends on the operand data (toggle rate) and on how many CUs are busy.  This probe issues nothing but
// register-resident MFMAs (no LDS, no memory; 16 independent accumulators per wave, 4 x 4 fragments) for ~10 ms per
// configuration and prints TFLOP/s for: 64 .. 256 workgroups (one per CU), 1 or 2 waves per SIMD, random fp16 operands
// in [-1, 1) or all-zero operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters,
                                                 int zero) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
  uint32_t s = seed[tid % 4096] * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      a[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
      s = s * 1664525u + 1013904223u;
      b[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = t;
}

// the accumulator layout of gemm16_quad_256x256_kernel: 64 accumulators (8 x 8 fragments) pinned in AGPRs by the asm form
// of the instruction, one wave per SIMD
typedef __attribute__((ext_vector_type(8))) short frag8;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void mfma_loop_agpr(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters, int zero) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 af[8], bf[8];
  uint32_t s = seed[tid % 4096] * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      af[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
      s = s * 1664525u + 1013904223u;
      bf[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
    }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[j]), "v"(af[i]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float t = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = t;
}


// the same experiment on v_mfma_f32_32x32x16_f16 (round 4): twice the flops per instruction and per operand-register read
// (8 + 8 operand registers feed 32768 flops instead of 16384), same nominal rate (one per 32 cycles and SIMD).  2 x 2
// fragments, 4 accumulators of 16 registers.
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void mfma_loop32(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters,
                                                   int zero) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
  uint32_t s = seed[tid % 4096] * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      a[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
      s = s * 1664525u + 1013904223u;
      b[i][e] = zero ? (_Float16)0.f : (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
    }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(r & 1) * 2 + i], b[(r >> 1 & 1) * 2 + j], acc[i][j], 0, 0, 0);
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
  printf("# register-resident v_mfma_f32_16x16x32_f16 only; TFLOP/s over the whole chip (nominal peak 2516 at 2.4 GHz)\n");
  printf("%6s %10s %8s %10s %10s\n", "CUs", "waves/SIMD", "data", "ms", "TFLOP/s");
  for (int zero = 0; zero < 2; ++zero)
    for (int wps = 1; wps <= 2; ++wps)
      for (int cus = 64; cus <= 256; cus += 64) {
        const int iters = 60000 / wps;               // 64 MFMAs per iteration and wave
        hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters / 8, zero);   // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters, zero);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 4 * wps * (double)iters * 64 * 16384.0;
        printf("%6d %10d %8s %10.3f %10.1f\n", cus, wps, zero ? "zeros" : "random", ms, flops / (ms * 1e-3) / 1e12);
      }
  printf("# 64 accumulators in AGPRs (asm form), 8 x 8 fragments, one wave per SIMD\n");
  for (int zero = 0; zero < 2; ++zero)
    for (int cus = 64; cus <= 256; cus += 64) {
      const int iters = 60000;
      hipLaunchKernelGGL(mfma_loop_agpr, dim3(cus), dim3(256), 0, 0, seed, out, iters / 8, zero);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_loop_agpr, dim3(cus), dim3(256), 0, 0, seed, out, iters, zero);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)cus * 4 * (double)iters * 64 * 16384.0;
      printf("%6d %10s %8s %10.3f %10.1f\n", cus, "1 (agpr)", zero ? "zeros" : "random", ms, flops / (ms * 1e-3) / 1e12);
    }
  printf("# register-resident v_mfma_f32_32x32x16_f16 only (2 x 2 fragments, 4 accumulators)\n");
  for (int zero = 0; zero < 2; ++zero)
    for (int wps = 1; wps <= 2; ++wps)
      for (int cus = 64; cus <= 256; cus += 64) {
        const int iters = 60000 / wps;               // 32 MFMAs of 32768 flops per iteration and wave
        hipLaunchKernelGGL(mfma_loop32, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters / 8, zero);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop32, dim3(cus), dim3(256 * wps), 0, 0, seed, out, iters, zero);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 4 * wps * (double)iters * 32 * 32768.0;
        printf("%6d %10d %8s %10.3f %10.1f\n", cus, wps, zero ? "zeros" : "random", ms, flops / (ms * 1e-3) / 1e12);
      }
  return 0;
}
