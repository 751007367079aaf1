// Shared device/host helpers for the gfx950 kernels of libw2v2hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/w2v2_hip.h"

// ---------------------------------------------------------------- error plumbing (never throw)
extern thread_local char g_w2v2_err[512];
#define W2V2_FAIL(...)                                   \
  do {                                                   \
    snprintf(g_w2v2_err, sizeof(g_w2v2_err), __VA_ARGS__); \
    return -1;                                           \
  } while (0)
#define W2V2_CHECK_LAUNCH(name)                                                       \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) W2V2_FAIL("%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)
#define W2V2_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) W2V2_FAIL(__VA_ARGS__); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- bf16 storage type
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round to nearest even on the gfx950 converter (v_cvt_pk_bf16_f32: one instruction per TWO values; the integer
// add/shift formulation cost ~6 VALU ops per value in every epilogue that stores bf16)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
  bf16x2_hw v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// 8-element vector access (16 B for bf16, 32 B for f32); p must be 16-byte aligned.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  float v[8];
  __device__ __forceinline__ void load(const float* p) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Vec8<bf16_t> {
  float v[8];
  __device__ __forceinline__ void load(const bf16_t* p) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x2_to_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 round-off class): 1 rcp + 1 exp + 7 fma
// instead of the ~40-instruction libm erff.  GELU is evaluated ~1e9 times per training step (conv layer 0,
// FFN1 epilogue, its backward), always as a serial tail of a kernel.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float y = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// ---------------------------------------------------------------- counter-based RNG (dropout)
// keep(seed, i) must be recomputable in the backward, so no mask is ever stored.  It sits in the inner loops of
// the attention kernels (T^2 decisions per head, three times per step) where it used to be the single largest
// cost: 32-bit integer multiplies run at quarter rate on CDNA (16 clk per wave instruction).  One murmur3
// finaliser (2 multiplies) over the PAIR index i >> 1 yields 32 bits = two 16-bit uniforms, for elements 2j and
// 2j + 1; the seed is mixed into a key on the scalar unit (wave-uniform).  p is quantised to 1/65536.
__device__ __forceinline__ uint32_t rng_key(uint64_t seed) {
  uint32_t k = (uint32_t)seed * 0x9E3779B1u ^ (uint32_t)(seed >> 32) * 0x85EBCA77u ^ 0x27D4EB2Fu;
  k ^= k >> 15; k *= 0x2C1B3C6Du;
  k ^= k >> 12; k *= 0x297A2D39u;
  k ^= k >> 15;
  return k;
}
__device__ __forceinline__ uint32_t rng_pair(uint32_t key, uint64_t pair_idx) {
  uint32_t x = (uint32_t)pair_idx ^ key ^ __umul24((uint32_t)(pair_idx >> 32), 0x9E3779u);
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)fminf(p * 65536.0f + 0.5f, 65535.0f); }
// returns 0 (dropped, probability p) or 1/(1-p)
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  const uint32_t h = rng_pair(rng_key(seed), idx >> 1);
  const uint32_t r = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return r >= drop_thr16(p) ? inv_keep : 0.0f;
}
// elements idx_even and idx_even + 1 from one hash (idx_even must be even)
__device__ __forceinline__ void drop_scale2(uint64_t seed, uint64_t idx_even, float p, float inv_keep, float& s0,
                                            float& s1) {
  const uint32_t h = rng_pair(rng_key(seed), idx_even >> 1);
  const uint32_t thr = drop_thr16(p);
  s0 = (h & 0xffffu) >= thr ? inv_keep : 0.0f;
  s1 = (h >> 16) >= thr ? inv_keep : 0.0f;
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
