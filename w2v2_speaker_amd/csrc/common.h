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

// ---------------------------------------------------------------- LDS-DMA piece (global_load_lds_dwordx4)
// 16 bytes per lane from `src` (per lane) to LDS at lds + 16 * lane (`lds`: wave-uniform pointer into __shared__ memory).
// Inline assembly instead of __builtin_amdgcn_global_load_lds ON PURPOSE: the compiler's waitcnt pass treats an LDS read
// behind the builtin as possibly aliasing the piece and puts its own `s_waitcnt vmcnt(0)` in front of the first ds_read
// that follows (round 6, found in the ISA of gemm_f32_dma.hip; the same wait stood at the top of the steady-state K loop
// of the phased and ring GEMMs, behind their own counted `vmcnt(8)` / `vmcnt(6)`): every K tile then waits for ALL
// pieces in flight, which is exactly what the counted waits of a multi-stage ring are there to avoid.  A kernel that uses
// this helper must order every piece against its readers itself (s_waitcnt vmcnt(n) + barrier) -- __syncthreads() does
// NOT wait for these pieces -- and must not use the builtin as well (the compiler does not know M0 changed here).
__device__ __forceinline__ void w2v2_dma16(const void* src, void* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(src) : "memory");
}

// `s_waitcnt vmcnt(0)` as an instruction the COMPILER sees (the builtin, not inline assembly).  Its waitcnt pass does not
// read the counted waits the kernels write in assembly: with vector-memory operations of its own still pending at a K
// loop's header (the previous tile's epilogue in a persistent kernel, the kernel-argument-dependent loads of the
// prologue) it guards the first register it overwrites INSIDE the loop with its own vmcnt(0) -- executed in every
// iteration, where it also waits for every LDS-DMA piece in flight.  One visible full wait in front of the loop (behind
// the prologue's pieces, which it also waits for) leaves the pass nothing to guard.
__device__ __forceinline__ void w2v2_vmcnt0_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt 0, expcnt 7, lgkmcnt 15

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

// ---------------------------------------------------------------- fp16 storage type
// IEEE half (11-bit significand: 8x the precision of bf16 at the same MFMA rate; the reference itself trains under
// fp16 AMP, config/experiment/speaker_wav2vec2_aam.yaml:17).  A distinct C++ type so templates can tell the two
// 16-bit activation formats apart; the backward runs under a loss scale (see optim.hip).
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_hw;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
__device__ __forceinline__ float f16_to_f32(f16_t v) { return (float)v; }
__device__ __forceinline__ f16_t f32_to_f16(float f) { return (f16_t)f; }
__device__ __forceinline__ uint32_t f32x2_to_f16x2(float lo, float hi) {
  f32x2_hw f;
  f[0] = lo;
  f[1] = hi;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_hw));   // v_cvt_pk_f16_f32 (RNE)
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return f16_to_f32(v); }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return f32_to_f16(v); }

// two f32 -> one packed dword of the 16-bit activation type T (low half = first value)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return f32x2_to_f16x2(lo, hi); }
// one packed dword -> two f32
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float& lo, float& hi) {
  const f16x2_hw h = __builtin_bit_cast(f16x2_hw, w);
  lo = (float)h[0];
  hi = (float)h[1];
}

// WRITE-THROUGH 16-byte store (global_store_dwordx4 ... sc1) for the once-written outputs of a kernel.  A dependent
// kernel boundary costs ~1.5-1.9 us + (bytes the predecessor left DIRTY in the eight L2s) / 6 TB/s
// (MI355X_MICROARCH.md, "boundary").  Written through, the bytes leave L2 while the kernel still computes and nothing is
// dirty at its end; the line is dropped from L2, which costs nothing here -- no kernel re-reads its own output.
// Measured (same-box ABAB of two library builds, tools/ab_bench.sh): 12.39 -> 12.34 ms/step (-0.4 %); the 4.7 us mean gap
// between kernels that tools/timeline_gaps.py shows under the tracer did NOT move (it is tracer + dispatch time: the
// un-traced step is only ~0.4 ms longer than the sum of its 254 kernels = the ~1.5 us floor per boundary).
// 16-byte stores cost the same either way; NARROWER sc1 stores are one fabric write each (2.7-12x per byte) and stay plain.
// (asm: the compiler's hazard recogniser does not see the store read its data registers -> s_nop 1.)
typedef __attribute__((ext_vector_type(4))) unsigned w2v2_u32x4;
__device__ __forceinline__ void store16_wt(void* p, uint4 v) {
  const w2v2_u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

// 8-element vector access (16 B for bf16, 32 B for f32); p must be 16-byte aligned.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  float v[8];
  __device__ __forceinline__ void load(const float* p) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    store16_wt(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    store16_wt(p + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
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
    store16_wt(p, make_uint4(w[0], w[1], w[2], w[3]));
  }
};

template <> struct Vec8<f16_t> {
  float v[8];
  __device__ __forceinline__ void load(const f16_t* p) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<f16_t>(w[i], v[2 * i], v[2 * i + 1]);
  }
  __device__ __forceinline__ void store(f16_t* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x2_to_f16x2(v[2 * i], v[2 * i + 1]);
    store16_wt(p, make_uint4(w[0], w[1], w[2], w[3]));
  }
};

// ---------------------------------------------------------------- matrix cores: one 16x16x32 step on raw 16-bit fragments
// Fragments travel as 8 x 16 raw bits (LDS images and registers do not care about the number format); the
// instruction is picked by the activation type: v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x32_f16 (same rate).
typedef __attribute__((ext_vector_type(8))) short frag8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_hw;
typedef __attribute__((ext_vector_type(4))) float f32x4_hw;
template <typename T> __device__ __forceinline__ f32x4_hw mfma16(frag8_t a, frag8_t b, f32x4_hw c);
template <> __device__ __forceinline__ f32x4_hw mfma16<bf16_t>(frag8_t a, frag8_t b, f32x4_hw c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0,
                                                 0, 0);
}
template <> __device__ __forceinline__ f32x4_hw mfma16<f16_t>(frag8_t a, frag8_t b, f32x4_hw c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0,
                                                0);
}

// The same instruction with the accumulator PINNED in AGPRs and updated in place (gemm16_quad_256x256_kernel: 256
// accumulator registers per wave = the whole AGPR file; left to the register allocator, the builtin form rotates the
// tuples through v_accvgpr moves, ~500 per K tile).  Plain asm (not volatile, no memory clobber): the compiler still
// schedules it, waits for the LDS reads that produce a / b, and keeps it in program order with its own accumulator only.
template <typename T> __device__ __forceinline__ void mfma16_agpr(frag8_t a, frag8_t b, f32x4_hw& c);
template <> __device__ __forceinline__ void mfma16_agpr<bf16_t>(frag8_t a, frag8_t b, f32x4_hw& c) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma16_agpr<f16_t>(frag8_t a, frag8_t b, f32x4_hw& c) {
  asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// acc + lo + hi of one packed pair, as ONE v_dot2c_f32_{bf16,f16} against a (1, 1) pair: column sums of fragments
// that are in registers anyway (bias gradients).  The (1, 1) operand must live in a VGPR: as a 32-bit literal
// (what the compiler emits for a constant) the packed-16 operand of v_dot2c is not read as two halves on gfx950 and
// the sums come out wrong -- ones_pair() therefore hides the constant behind an empty asm.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_dot_t;
template <typename T> __device__ __forceinline__ uint32_t ones_pair();
template <> __device__ __forceinline__ uint32_t ones_pair<bf16_t>() {
  uint32_t w = 0x3f803f80u;
  asm volatile("" : "+v"(w));
  return w;
}
template <> __device__ __forceinline__ uint32_t ones_pair<f16_t>() {
  uint32_t w = 0x3c003c00u;
  asm volatile("" : "+v"(w));
  return w;
}
template <typename T> __device__ __forceinline__ float pair_sum_add(uint32_t w, uint32_t ones, float acc);
template <> __device__ __forceinline__ float pair_sum_add<bf16_t>(uint32_t w, uint32_t ones, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_dot_t, w), __builtin_bit_cast(bf16x2_dot_t, ones), acc,
                                         false);
}
template <> __device__ __forceinline__ float pair_sum_add<f16_t>(uint32_t w, uint32_t ones, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_hw, w), __builtin_bit_cast(f16x2_hw, ones), acc, false);
}

// ---------------------------------------------------------------- full-line stores of MFMA-layout tiles
// After a 16x16 MFMA a lane (c = lane & 15, q = lane >> 4) holds 16 consecutive output columns of row c as two 16-byte
// halves P0 | P1; the four q-lanes of a row cover 128 contiguous bytes.  Stored as they are, one wave instruction writes
// four scattered 16-byte pieces per row (4.6 TB/s chip-wide, tools/probes/store_pattern_probe.hip); when lanes c and
// c ^ 8 swap one half each (v_mov_dpp row_ror:8) every instruction writes 8 rows x 128 contiguous bytes (5.8 TB/s):
// lane c < 8 stores columns +0..7 of rows (c & 7) and (c & 7) + 8, lane c >= 8 columns +8..15 of the same two rows.
__device__ __forceinline__ uint32_t dpp_ror8(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint4 dpp_ror8(uint4 v) {
  return make_uint4(dpp_ror8(v.x), dpp_ror8(v.y), dpp_ror8(v.z), dpp_ror8(v.w));
}
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) {
  return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
// own halves (P0, P1) -> what this lane stores for rows (c & 7) and (c & 7) + 8
__device__ __forceinline__ void halves_to_lines(bool lo, uint4 p0, uint4 p1, uint4& da, uint4& db) {
  const uint4 y = dpp_ror8(sel4(lo, p1, p0));
  da = sel4(lo, p0, y);
  db = sel4(lo, y, p1);
}
// what this lane loaded from rows (c & 7) and (c & 7) + 8 -> its own halves (the same swap, inverted)
__device__ __forceinline__ void lines_to_halves(bool lo, uint4 la, uint4 lb, uint4& p0, uint4& p1) {
  const uint4 z = dpp_ror8(sel4(lo, lb, la));
  p0 = sel4(lo, la, z);
  p1 = sel4(lo, z, lb);
}
template <typename TC> __device__ __forceinline__ void unpack8(uint4 w, float* v) {
  unpack2<TC>(w.x, v[0], v[1]); unpack2<TC>(w.y, v[2], v[3]); unpack2<TC>(w.z, v[4], v[5]); unpack2<TC>(w.w, v[6], v[7]);
}
template <typename TC> __device__ __forceinline__ uint4 pack8(const float* v) {
  return make_uint4(pack2<TC>(v[0], v[1]), pack2<TC>(v[2], v[3]), pack2<TC>(v[4], v[5]), pack2<TC>(v[6], v[7]));
}

// ---------------------------------------------------------------- activation-dtype dispatch of the C entry points
// `AT` names the storage type inside the statement; unknown codes fail with the entry point's name.
#define W2V2_DISPATCH_ACT(DT, NAME, ...)                                       \
  switch (DT) {                                                                \
    case W2V2_BF16: { using AT = bf16_t; __VA_ARGS__; } break;                  \
    case W2V2_F16: { using AT = f16_t; __VA_ARGS__; } break;                    \
    case W2V2_F32: { using AT = float; __VA_ARGS__; } break;                    \
    default: W2V2_FAIL("%s: bad dtype %d", NAME, (int)(DT));                   \
  }
// the two 16-bit formats only (kernels that need the matrix cores)
#define W2V2_DISPATCH_16(DT, NAME, ...)                                        \
  switch (DT) {                                                                \
    case W2V2_BF16: { using AT = bf16_t; __VA_ARGS__; } break;                  \
    case W2V2_F16: { using AT = f16_t; __VA_ARGS__; } break;                    \
    default: W2V2_FAIL("%s: needs a 16-bit activation dtype (got %d)", NAME, (int)(DT)); \
  }

// ---------------------------------------------------------------- math
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 round-off class): 1 rcp + 1 exp + 7 fma
// (the reciprocal is the hardware's v_rcp_f32, 1 ulp: `__frcp_rn` -- correctly rounded -- expands to the ten-instruction
// IEEE division sequence, a third of the whole GELU epilogue)
// instead of the ~40-instruction libm erff.  GELU is evaluated ~1e9 times per training step (conv layer 0,
// FFN1 epilogue, its backward), always as a serial tail of a kernel.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
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

// GELU and its derivative from ONE evaluation of the erf polynomial and ONE exponential: exp(-(x / sqrt 2)^2) is both the
// tail of erf and, times 1 / sqrt(2 pi), the normal density.  y is bit-identical to gelu_f(x).
__device__ __forceinline__ void gelu_both_f(float x, float& y, float& dy) {
  const float xs = x * 0.70710678118654752f;
  const float ax = fabsf(xs);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float ex = __expf(-ax * ax);
  const float erfv = copysignf(1.0f - poly * t * ex, xs);
  y = 0.5f * x * (1.0f + erfv);
  dy = fmaf(x * 0.39894228040143268f, ex, 0.5f * (1.0f + erfv));
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
