// dma_issue_probe -- what does ONE LDS-DMA piece (16 bytes per lane, 1 KiB per wave-instruction) cost the matrix pipe of
// the SIMD that issues it, and does the addressing form matter?   (build: hipcc --offload-arch=gfx950 -O3)
//
// 256 workgroups x 512 threads (two waves per SIMD, as the 256x128 ring GEMM).  Every wave runs `iters` rounds of 32
// register-resident v_mfma_f32_16x16x32_f16 and issues P pieces per round, evenly spread between the MFMAs, from an
// L2-resident 8 MiB buffer into its own LDS slots.  Printed: time per round for P = 0, 2, 4, 6, 8 and the extra time per
// piece, for
//   V0  global_load_lds_dwordx4 v[addr : addr + 1], off             (64-bit per-lane address: what the kernels use)
//   V1  global_load_lds_dwordx4 v_offset, s[base : base + 1]        (SGPR base + 32-bit per-lane offset)
//   V2  buffer_load_dwordx4 v_offset, s[rsrc : rsrc + 3], 0 offen lds
//   V4  global_load_lds_dword v[addr : addr + 1], off               (4 bytes per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <utility>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int V, int P>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, float* __restrict__ out, int iters, i32x4 rsrc) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f16x8 a[4], b[4];
  uint32_t s = (blockIdx.x * 512 + tid) * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      a[i][e] = (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
      s = s * 1664525u + 1013904223u;
      b[i][e] = (_Float16)((float)(int)(s >> 8 & 0xffff) / 32768.f - 1.f);
    }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane source: 16 bytes, rows of 128 bytes as the GEMM tiles have them; the round index moves it through 8 MiB
  uint32_t off = ((blockIdx.x * 8 + wave) * 8192u + (lane >> 3) * 128u + (lane & 7) * 16u) & ((8u << 20) - 1);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    sfor<32>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      acc[(j >> 2) & 3][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(j >> 2) & 3], b[j & 3], acc[(j >> 2) & 3][j & 3], 0, 0, 0);
      if constexpr (P > 0) {
        if constexpr ((j + 1) % (32 / P) == 0) {
          constexpr int p = (j + 1) / (32 / P) - 1;
          const uint32_t o = (off + p * 1024u) & ((8u << 20) - 1);
          const uint32_t l = lds0 + p * 1024;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (V == 0) {
            const char* ptr = src + o;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(ptr) : "memory");
          } else if constexpr (V == 1) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(o), "s"(src) : "memory");
          } else if constexpr (V == 2) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l), "v"(o), "s"(rsrc) : "memory");
          } else {
            const char* ptr = src + o;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(l), "v"(ptr) : "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    off = (off + 65536u) & ((8u << 20) - 1);
    if (P > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = t + smem[tid];
}

template <int V, int P>
static double run(const char* src, float* out, int iters, i32x4 rsrc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((probe<V, P>), dim3(256), dim3(512), 65536, 0, src, out, iters / 8, rsrc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V, P>), dim3(256), dim3(512), 65536, 0, src, out, iters, rsrc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)ms * 1e6 / iters;      // ns per round
}

template <int V>
static void variant(const char* name, const char* src, float* out, int iters, i32x4 rsrc) {
  const double t0 = run<V, 0>(src, out, iters, rsrc);
  const double t2 = run<V, 2>(src, out, iters, rsrc);
  const double t4 = run<V, 4>(src, out, iters, rsrc);
  const double t8 = run<V, 8>(src, out, iters, rsrc);
  // a round = 32 MFMAs per wave, two waves per SIMD: 64 x 16 = 1024 pipe cycles at full rate
  printf("%-46s ns/round P=0 %7.1f  P=2 %7.1f  P=4 %7.1f  P=8 %7.1f   extra per piece and SIMD (two waves each issue P): "
         "%.1f / %.1f / %.1f ns\n", name, t0, t2, t4, t8, (t2 - t0) / 4, (t4 - t0) / 8, (t8 - t0) / 16);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "0124";
  char* src;
  float* out;
  hipMalloc(&src, 8 << 20);
  hipMemset(src, 0, 8 << 20);
  hipMalloc(&out, 256 * 512 * 4);
  const uint64_t base = (uint64_t)(uintptr_t)src;
  i32x4 rsrc = {(int)(base & 0xffffffffu), (int)((base >> 32) & 0xffff), (int)(8 << 20), 0x00020000};
  const int iters = 20000;
  if (strchr(which, '0')) variant<0>("V0 global_load_lds_dwordx4 vaddr64", src, out, iters, rsrc);
  if (strchr(which, '1')) variant<1>("V1 global_load_lds_dwordx4 voffset32 + saddr", src, out, iters, rsrc);
  if (strchr(which, '2')) variant<2>("V2 buffer_load_dwordx4 offen lds", src, out, iters, rsrc);
  if (strchr(which, '4')) variant<4>("V4 global_load_lds_dword (4 B per lane)", src, out, iters, rsrc);
  return 0;
}
