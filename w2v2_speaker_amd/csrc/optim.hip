// optim.hip -- fused Adam over one flat f32 parameter arena (torch.optim.Adam semantics,
// ref: config/optim/algo/adam.yaml:1-16, wired at src/main.py:323-335).  One launch updates every
// trainable parameter and refreshes the bf16 copy the MFMA GEMMs read.  HBM-bound:
// 4 f32 streams read (p, g, m, v) + 3 written (p, m, v) + 2 B/param bf16 copy = 30 B/param.
#include "common.h"
#include <stdlib.h>

template <typename TB, int U>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   TB* __restrict__ pb, int64_t n, float lr, float b1, float b2,
                                                   float eps, float step_size, float inv_sqrt_bc2, float gscale,
                                                   const float* __restrict__ scaler, int step, int skip_slot) {
  // dynamic loss scaling (fp16 activations): gradients carry the factor scaler[0]; a step whose gradients held a
  // non-finite value (scaler[1] != 0, set by grad_scaler_check) is skipped as a whole, like torch's GradScaler.step
  if (scaler != nullptr) {
    if (scaler[1] != 0.f) return;
    gscale /= scaler[0];
    // torch's GradScaler skips optimizer.step() on overflow, so Adam's per-parameter step count does NOT advance on a
    // skipped step.  The host counts every call (it never reads the device record); scaler[skip_slot] counts the skipped
    // ones of this parameter range, so the bias corrections are rebuilt here from t = step - skipped.
    if (skip_slot > 0) {
      const float t = fmaxf((float)step - scaler[skip_slot], 1.0f);
      // double pow: at small t, 1 - 0.999^t ~ 1e-3 and the ~1-ulp error of the f32 exp2 / log2 path would be 5e-5..1e-4
      // relative in the correction; once per thread, invisible next to the 30 B/parameter stream.  The
      // host's bias_corr1 / bias_corr2 arguments are NOT used on this path.
      step_size = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
      inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
    }
  }
  const int64_t nv = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // U independent 16-byte vectors per thread and pass: 4 U loads in flight per lane before the first use
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < nv; i0 += stride * U) {
    f32x4_hw pp[U], gg[U], mm[U], vv[U];
    // every array is touched once per step: streaming (non-temporal) accesses keep the 30 B/parameter out of the way
    // of the L2 / Infinity Cache contents the next forward wants (the 16-bit operand copy written below)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nv) {
        pp[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_hw*>(p) + i);
        gg[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_hw*>(g) + i);
        mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_hw*>(m) + i);
        vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_hw*>(v) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= nv) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gg[u][e] * gscale;
        mm[u][e] = b1 * mm[u][e] + (1.0f - b1) * gr;
        vv[u][e] = b2 * vv[u][e] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vv[u][e]) * inv_sqrt_bc2 + eps;
        pp[u][e] -= step_size * mm[u][e] / denom;
      }
      __builtin_nontemporal_store(pp[u], reinterpret_cast<f32x4_hw*>(p) + i);
      __builtin_nontemporal_store(mm[u], reinterpret_cast<f32x4_hw*>(m) + i);
      __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4_hw*>(v) + i);
      if (pb != nullptr) {
        uint2 w;
        w.x = pack2<TB>(pp[u][0], pp[u][1]);
        w.y = pack2<TB>(pp[u][2], pp[u][3]);
        reinterpret_cast<uint2*>(pb)[i] = w;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nv << 2) + threadIdx.x;
    const float gr = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gr;
    const float vi = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    const float pn = p[i] - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pn;
    if (pb != nullptr) pb[i] = from_f32<TB>(pn);
  }
}

// ------------------------------------------------------------------------------ dynamic loss scaling (fp16 activations)
// The reference trains under PL `precision: 16` = torch.cuda.amp.GradScaler (config/experiment/
// speaker_wav2vec2_aam.yaml:17): the loss is multiplied by `scale` before backward so that fp16 activation
// gradients stay in range; a step with a non-finite gradient is skipped and the scale halved, and after
// `growth_interval` clean steps it is doubled.  Everything lives in a 4-float device record, so a training step
// never synchronises with the host:  state = {scale, found_inf, growth_tracker, skipped_steps}.
__global__ __launch_bounds__(256) void scaler_check_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ state) {
  const int64_t nv = n >> 2;
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    // finite <=> |x| <= FLT_MAX; the sum of absolute values is inf or NaN iff any element is
    const float a = fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
    bad |= !(a <= 3.402823466e38f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !(fabsf(g[(nv << 2) + threadIdx.x]) <= 3.402823466e38f);
  if (__any(bad) && (threadIdx.x & 63) == 0) state[1] = 1.0f;      // benign race: every writer stores the same value
}
__global__ void scaler_update_kernel(float* __restrict__ state, float growth, float backoff, int interval, int ranges) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[1] != 0.f) {
    state[0] = fmaxf(state[0] * backoff, 1.0f);
    state[2] = 0.f;
    state[3] += 1.0f;
    if (ranges & 1) state[4] += 1.0f;      // skipped optimiser steps per parameter range (8-float records only)
    if (ranges & 2) state[5] += 1.0f;
  } else {
    state[2] += 1.0f;
    if (state[2] >= (float)interval) {
      state[0] = fminf(state[0] * growth, 16777216.0f);
      state[2] = 0.f;
    }
  }
  state[1] = 0.f;
}

// residual plane of the two-term 16-bit weights: lo = T(p - T(p)) over a table of (offset, count) ranges
template <typename TB>
__global__ __launch_bounds__(256) void weight_residual_kernel(const float* __restrict__ p, TB* __restrict__ lo,
                                                              const int64_t* __restrict__ table) {
  const int64_t off = table[2 * blockIdx.y], n = table[2 * blockIdx.y + 1];
  const int64_t nv = n >> 2;              // ranges are 64-element aligned in the arena; the tail loop covers any rest
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(p + off)[i];
    const float a[4] = {v.x, v.y, v.z, v.w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = a[e] - to_f32<TB>(from_f32<TB>(a[e]));
    uint2 w;
    w.x = pack2<TB>(r[0], r[1]);
    w.y = pack2<TB>(r[2], r[3]);
    reinterpret_cast<uint2*>(lo + off)[i] = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = off + (nv << 2) + threadIdx.x;
    lo[i] = from_f32<TB>(p[i] - to_f32<TB>(from_f32<TB>(p[i])));
  }
}

extern "C" int w2v2_weight_residual(const float* p, void* lo, const int64_t* table, int n_ranges, int dtype,
                                    void* stream) {
  W2V2_REQUIRE(p && lo && table && n_ranges >= 0, "weight_residual: bad arguments");
  if (n_ranges == 0) return 0;
  W2V2_DISPATCH_16(dtype, "weight_residual",
    hipLaunchKernelGGL(weight_residual_kernel<AT>, dim3(64, n_ranges), dim3(256), 0, as_stream(stream), p, (AT*)lo,
                       table););
  W2V2_CHECK_LAUNCH("weight_residual");
  return 0;
}

extern "C" int w2v2_grad_scaler_check(const float* g, int64_t n, float* state, void* stream) {
  W2V2_REQUIRE(g && state && n >= 0, "grad_scaler_check: bad arguments");
  if (n == 0) return 0;
  int64_t nb = cdiv(n >> 2, 256 * 8);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(scaler_check_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), g, n, state);
  W2V2_CHECK_LAUNCH("grad_scaler_check");
  return 0;
}

extern "C" int w2v2_grad_scaler_update(float* state, float growth, float backoff, int growth_interval,
                                       int skipped_ranges, void* stream) {
  W2V2_REQUIRE(state && growth >= 1.f && backoff > 0.f && backoff <= 1.f && growth_interval > 0 &&
                   skipped_ranges >= 0 && skipped_ranges <= 3, "grad_scaler_update: bad arguments");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), state, growth, backoff,
                     growth_interval, skipped_ranges);
  W2V2_CHECK_LAUNCH("grad_scaler_update");
  return 0;
}

extern "C" int w2v2_adam_step(float* p, const float* g, float* m, float* v, void* pb, int pb_dtype, int64_t n,
                              float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                              float grad_scale, const float* scaler_state, int step, int skip_slot, void* stream) {
  W2V2_REQUIRE(p && g && m && v && n >= 0, "adam_step: bad arguments");
  W2V2_REQUIRE(skip_slot == 0 || ((skip_slot == 4 || skip_slot == 5) && scaler_state != nullptr && step >= 1),
               "adam_step: skip_slot must be 0, or 4 / 5 with an 8-float scaler record and step >= 1");
  W2V2_REQUIRE(bias_corr1 > 0.f && bias_corr2 > 0.f, "adam_step: bias corrections must be > 0");
  if (n == 0) return 0;
  // A/B knobs (tools): W2V2_ADAM_U = vectors per thread and pass (1, 2, 4), W2V2_ADAM_BLOCKS = grid cap
  static const int env_u_raw = getenv("W2V2_ADAM_U") ? atoi(getenv("W2V2_ADAM_U")) : 2;   // 2: 495 vs 549 us over the 99.4 M-parameter arena
  static const int env_u = env_u_raw >= 4 ? 4 : (env_u_raw >= 2 ? 2 : 1);
  // grid cap: none by default (round 6).  The 8192-block cap of round 3 made every thread walk 6 (w2v2-base) to 20
  // (wav2vec2-large) strides of the arena; one pass of one or two vectors per thread measures 456 vs 561 us over the base
  // arena (6.5 vs 5.3 TB/s, tools/adam_sweep.sh, profiles/r06_adam_sweep.txt) and -0.7 % of the step (ABAB x 4).
  static const int env_nb = getenv("W2V2_ADAM_BLOCKS") ? atoi(getenv("W2V2_ADAM_BLOCKS")) : (1 << 20);
  int64_t nb = cdiv(n >> 2, 256 * env_u);
  if (nb > env_nb) nb = env_nb;
  if (nb < 1) nb = 1;
  if (pb == nullptr) pb_dtype = W2V2_BF16;
#define W2V2_ADAM_LAUNCH(U_)                                                                                        \
  hipLaunchKernelGGL((adam_kernel<AT, U_>), dim3((unsigned)nb), dim3(256), 0, as_stream(stream), p, g, m, v, (AT*)pb, n, \
                     lr, beta1, beta2, eps, lr / bias_corr1, 1.0f / sqrtf(bias_corr2), grad_scale, scaler_state, step, skip_slot)
  W2V2_DISPATCH_16(pb_dtype, "adam_step",
    if (env_u == 4) W2V2_ADAM_LAUNCH(4); else if (env_u == 2) W2V2_ADAM_LAUNCH(2); else W2V2_ADAM_LAUNCH(1););
#undef W2V2_ADAM_LAUNCH
  W2V2_CHECK_LAUNCH("adam_step");
  return 0;
}
