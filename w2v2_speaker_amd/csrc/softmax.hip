// softmax.hip -- row softmax (+ dropout on the probabilities) of the unfused attention path
// (HF:438-463 eager_attention_forward).  One wave per score row; used for sequence lengths the
// fused kernel (attention.hip) does not cover (full-length test utterances, T up to ~7k frames).
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, T* __restrict__ p,
                                                          T* __restrict__ pd, int64_t rows, int Tn, int64_t ld,
                                                          float dp, float inv_keep, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < Tn; c += 64) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < Tn; c += 64) sum += __expf(sr[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < Tn; c += 64) {
    const float pv = __expf(sr[c] - mx) * inv;
    p[row * ld + c] = from_f32<T>(pv);
    if (pd != nullptr) pd[row * ld + c] = from_f32<T>(pv * drop_scale(seed, (uint64_t)(row * ld + c), dp, inv_keep));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dpd, const T* __restrict__ p,
                                                          T* __restrict__ ds, int64_t rows, int Tn, int64_t ld,
                                                          float dp, float inv_keep, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < Tn; c += 64) {
    const int64_t i = row * ld + c;
    float g = dpd[i];
    if (dp > 0.f) g *= drop_scale(seed, (uint64_t)i, dp, inv_keep);
    dot += g * to_f32<T>(p[i]);
  }
  dot = wave_sum(dot);
  for (int c = lane; c < Tn; c += 64) {
    const int64_t i = row * ld + c;
    float g = dpd[i];
    if (dp > 0.f) g *= drop_scale(seed, (uint64_t)i, dp, inv_keep);
    ds[i] = from_f32<T>(to_f32<T>(p[i]) * (g - dot));
  }
}

extern "C" int w2v2_softmax_fwd(const float* s, void* p, void* p_drop, int64_t rows, int T, int64_t ld, float drop_p,
                                uint64_t seed, int dtype, void* stream) {
  W2V2_REQUIRE(s && p && rows >= 0 && T > 0 && ld >= T && drop_p >= 0.f && drop_p < 1.f, "softmax_fwd: bad arguments");
  if (rows == 0) return 0;
  if (drop_p <= 0.f) p_drop = nullptr;
  else W2V2_REQUIRE(p_drop != nullptr, "softmax_fwd: drop_p > 0 needs p_drop");
  const float ik = 1.0f / (1.0f - drop_p);
  dim3 grid((unsigned)cdiv(rows, 4));
  W2V2_DISPATCH_ACT(dtype, "softmax_fwd",
    hipLaunchKernelGGL(softmax_fwd_kernel<AT>, grid, dim3(256), 0, as_stream(stream), s, (AT*)p,
                       (AT*)p_drop, rows, T, ld, drop_p, ik, seed););
  W2V2_CHECK_LAUNCH("softmax_fwd");
  return 0;
}

extern "C" int w2v2_softmax_bwd(const float* dp_drop, const void* p, void* ds, int64_t rows, int T, int64_t ld,
                                float drop_p, uint64_t seed, int dtype, void* stream) {
  W2V2_REQUIRE(dp_drop && p && ds && rows >= 0 && T > 0 && ld >= T && drop_p >= 0.f && drop_p < 1.f,
               "softmax_bwd: bad arguments");
  if (rows == 0) return 0;
  const float ik = 1.0f / (1.0f - drop_p);
  dim3 grid((unsigned)cdiv(rows, 4));
  W2V2_DISPATCH_ACT(dtype, "softmax_bwd",
    hipLaunchKernelGGL(softmax_bwd_kernel<AT>, grid, dim3(256), 0, as_stream(stream), dp_drop, (const AT*)p,
                       (AT*)ds, rows, T, ld, drop_p, ik, seed););
  W2V2_CHECK_LAUNCH("softmax_bwd");
  return 0;
}
