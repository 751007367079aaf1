// pool.hip -- temporal pooling of the encoder output (ref: src/layers/pooling.py:24-44,74-80,118-136).
// x [B,T,H] channels-last -> f32 embedding.  mean+std: a workgroup owns (utterance, 128 channels);
// 16 column lanes x 16-byte vectors give fully coalesced 256-B row reads, the 16 time lanes
// (4 lane groups x 4 waves) are folded with wave shuffles (xor 16, 32) and one LDS hop across waves.
// Two passes over the (L2-resident) slab: mean, then centred sum of squares -> torch.std_mean parity.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void pool_meanstd_kernel(const T* __restrict__ x, float* __restrict__ out, int Tn,
                                                           int H, int with_std) {
  __shared__ float red[4][16][8];
  __shared__ float meanv[16][8];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane & 15, tl = (lane >> 4) + 4 * wave;  // column lane, time lane (0..15)
  const int col = (blockIdx.x * 16 + cl) * 8;
  const bool active = col < H;
  const T* xb = x + (int64_t)b * Tn * H + col;
  float acc[8] = {};
  if (active)
    for (int t = tl; t < Tn; t += 16) {
      Vec8<T> v;
      v.load(xb + (int64_t)t * H);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v.v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 16, 64);
    acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  if (lane < 16)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][cl][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 3, e = threadIdx.x & 7;
    meanv[c][e] = (red[0][c][e] + red[1][c][e] + red[2][c][e] + red[3][c][e]) / (float)Tn;
  }
  __syncthreads();
  float mu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mu[e] = meanv[cl][e];
  const int mean_off = with_std ? H : 0;
  const int ostride = with_std ? 2 * H : H;
  if (active && wave == 0 && lane < 16)
#pragma unroll
    for (int e = 0; e < 8; ++e) out[(int64_t)b * ostride + mean_off + col + e] = mu[e];
  if (!with_std) return;
  float sq[8] = {};
  if (active)
    for (int t = tl; t < Tn; t += 16) {
      Vec8<T> v;
      v.load(xb + (int64_t)t * H);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v.v[e] - mu[e]; sq[e] = fmaf(d, d, sq[e]); }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sq[e] += __shfl_xor(sq[e], 16, 64);
    sq[e] += __shfl_xor(sq[e], 32, 64);
  }
  __syncthreads();
  if (lane < 16)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][cl][e] = sq[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 3, e = threadIdx.x & 7;
    const int cc = (blockIdx.x * 16 + c) * 8 + e;
    if (cc < H) {
      const float m2 = red[0][c][e] + red[1][c][e] + red[2][c][e] + red[3][c][e];
      out[(int64_t)b * ostride + cc] = sqrtf(m2 / (float)(Tn - 1));  // unbiased; T == 1 -> NaN like torch
    }
  }
}

// dx = dmean/T + dstd * (x - mean) / ((T-1) * std)
template <typename T>
__global__ void pool_meanstd_bwd_kernel(const T* __restrict__ x, const float* __restrict__ out,
                                        const float* __restrict__ dout, T* __restrict__ dx, int B, int Tn, int H,
                                        int with_std) {
  const int nch = H >> 3;
  const int64_t total = (int64_t)B * Tn * nch;
  const int ostride = with_std ? 2 * H : H, mean_off = with_std ? H : 0;
  const float invT = 1.0f / (float)Tn, invT1 = 1.0f / (float)(Tn - 1);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int b = (int)(row / Tn);
    const int col = ch * 8;
    Vec8<T> o;
    if (with_std) {
      Vec8<T> v;
      v.load(x + row * H + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sd = out[(int64_t)b * ostride + col + e], mu = out[(int64_t)b * ostride + H + col + e];
        o.v[e] = dout[(int64_t)b * ostride + mean_off + col + e] * invT +
                 dout[(int64_t)b * ostride + col + e] * (v.v[e] - mu) * invT1 / sd;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = dout[(int64_t)b * ostride + col + e] * invT;
    }
    o.store(dx + row * H + col);
  }
}

// max / first / last: one thread per (b, c); coalesced over c.
template <typename T>
__global__ void pool_select_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int Tn, int H, int mode) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), c = (int)(i - (int64_t)b * H);
  const T* xb = x + (int64_t)b * Tn * H + c;
  float r;
  if (mode == 2) {
    r = to_f32<T>(xb[0]);
    for (int t = 1; t < Tn; ++t) r = fmaxf(r, to_f32<T>(xb[(int64_t)t * H]));
  } else if (mode == 3) {
    r = to_f32<T>(xb[0]);
  } else if (mode >= 16) {                  // explicit frame index (IndexPool1D "random": the host draws it)
    r = to_f32<T>(xb[(int64_t)(mode - 16) * H]);
  } else {
    r = to_f32<T>(xb[(int64_t)(Tn - 1) * H]);
  }
  out[i] = r;
}

template <typename T>
__global__ void pool_select_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dout, T* __restrict__ dx,
                                       int B, int Tn, int H, int mode) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), c = (int)(i - (int64_t)b * H);
  const T* xb = x + (int64_t)b * Tn * H + c;
  T* db = dx + (int64_t)b * Tn * H + c;
  int sel = 0;
  if (mode == 2) {
    float r = to_f32<T>(xb[0]);
    for (int t = 1; t < Tn; ++t) {
      const float v = to_f32<T>(xb[(int64_t)t * H]);
      if (v > r) { r = v; sel = t; }   // first arg-max, like torch.max(dim)
    }
  } else if (mode == 4) {
    sel = Tn - 1;
  } else if (mode >= 16) {
    sel = mode - 16;
  }
  for (int t = 0; t < Tn; ++t) db[(int64_t)t * H] = from_f32<T>(t == sel ? dout[i] : 0.f);
}

// ------------------------------------------------------------------------------------------ quantile pooling
// ref: src/layers/pooling.py:51-67 -- torch.quantile(x, [0, .25, .5, .75, 1], dim=time) (linear interpolation between
// the order statistics floor / ceil of q (T-1)), stacked as [B, 5 H] (quantile-major).  One thread per (b, channel),
// coalesced over channels.  The k-th smallest value is found WITHOUT sorting, by fixing the bits of its order-preserving
// integer key from the top (one counting pass over the T values per bit, all eight ranks of the five quantiles in the
// same pass): NB = 16 / 19 / 32 passes for bf16 / fp16 / f32 values, O(T) memory traffic from an L1/L2-resident column
// tile, any T.  Ties are ordered by time index (a stable sort), which is where torch's backward sends the gradient.
template <typename T> struct KeyBits;
template <> struct KeyBits<float> { static constexpr int NB = 32; };
template <> struct KeyBits<bf16_t> { static constexpr int NB = 16; };
template <> struct KeyBits<f16_t> { static constexpr int NB = 19; };     // f32 image of a half: 1 + 8 + 10 bits

__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int QN = 8;   // ranks: 0, lo/hi of the three inner quantiles, T-1
struct QRanks { int k[QN]; float w[3]; };
__device__ __forceinline__ QRanks quantile_ranks(int Tn) {
  QRanks r;
  r.k[0] = 0;
  r.k[7] = Tn - 1;
  const float qs[3] = {0.25f, 0.5f, 0.75f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float pos = qs[i] * (float)(Tn - 1);          // torch: rank = q * (n - 1) in f32
    const float lo = floorf(pos);
    r.k[1 + 2 * i] = (int)lo;
    r.k[2 + 2 * i] = (int)ceilf(pos);
    r.w[i] = pos - lo;
  }
  return r;
}
// keys of the QN order statistics of column xb[t * H], t < Tn (top NB bits of the 32-bit order key)
template <typename T>
__device__ __forceinline__ void select_keys(const T* __restrict__ xb, int Tn, int H, const int (&rank)[QN],
                                            uint32_t (&key)[QN]) {
  constexpr int NB = KeyBits<T>::NB;
  int rem[QN];
#pragma unroll
  for (int i = 0; i < QN; ++i) { key[i] = 0; rem[i] = rank[i]; }
  for (int bit = 31; bit >= 32 - NB; --bit) {
    int c0[QN];
#pragma unroll
    for (int i = 0; i < QN; ++i) c0[i] = 0;
    for (int t = 0; t < Tn; ++t) {
      const uint32_t k = order_key(to_f32<T>(xb[(int64_t)t * H])) >> bit;      // bits above `bit` + the bit itself
#pragma unroll
      for (int i = 0; i < QN; ++i) c0[i] += (k == (key[i] >> bit)) ? 1 : 0;    // prefix matches and this bit is 0
    }
#pragma unroll
    for (int i = 0; i < QN; ++i)
      if (rem[i] >= c0[i]) { rem[i] -= c0[i]; key[i] |= 1u << bit; }
  }
  // the undecided low bits are zero in the VALUE for these formats, i.e. all ones in the key of a negative number
#pragma unroll
  for (int i = 0; i < QN; ++i)
    if (!(key[i] & 0x80000000u)) key[i] |= (uint32_t)((1ull << (32 - NB)) - 1);
}
__device__ __forceinline__ float lerp_torch(float a, float b, float w) {
  return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);       // at::lerp
}

template <typename T>
__global__ void pool_quantile_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int Tn, int H) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), c = (int)(i - (int64_t)b * H);
  const T* xb = x + (int64_t)b * Tn * H + c;
  const QRanks r = quantile_ranks(Tn);
  uint32_t key[QN];
  select_keys<T>(xb, Tn, H, r.k, key);
  float* ob = out + (int64_t)b * 5 * H + c;
  ob[0] = key_value(key[0]);
#pragma unroll
  for (int q = 0; q < 3; ++q)
    ob[(int64_t)(q + 1) * H] = lerp_torch(key_value(key[1 + 2 * q]), key_value(key[2 + 2 * q]), r.w[q]);
  ob[(int64_t)4 * H] = key_value(key[7]);
}

template <typename T>
__global__ void pool_quantile_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dout, T* __restrict__ dx,
                                         int B, int Tn, int H) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), c = (int)(i - (int64_t)b * H);
  const T* xb = x + (int64_t)b * Tn * H + c;
  T* db = dx + (int64_t)b * Tn * H + c;
  const QRanks r = quantile_ranks(Tn);
  uint32_t key[QN];
  select_keys<T>(xb, Tn, H, r.k, key);
  // time index of each order statistic: among equal values the (rank - #smaller)-th occurrence (stable order)
  constexpr int SH = 32 - KeyBits<T>::NB;
  int less[QN], idx[QN];
#pragma unroll
  for (int j = 0; j < QN; ++j) { less[j] = 0; idx[j] = -1; }
  for (int t = 0; t < Tn; ++t) {
    const uint32_t k = order_key(to_f32<T>(xb[(int64_t)t * H])) >> SH;
#pragma unroll
    for (int j = 0; j < QN; ++j) less[j] += (k < (key[j] >> SH)) ? 1 : 0;
  }
  int seen[QN];
#pragma unroll
  for (int j = 0; j < QN; ++j) seen[j] = 0;
  for (int t = 0; t < Tn; ++t) {
    const uint32_t k = order_key(to_f32<T>(xb[(int64_t)t * H])) >> SH;
#pragma unroll
    for (int j = 0; j < QN; ++j)
      if (k == (key[j] >> SH)) {
        if (seen[j] == r.k[j] - less[j]) idx[j] = t;
        ++seen[j];
      }
  }
  const float* gb = dout + (int64_t)b * 5 * H + c;
  float coef[QN];
  coef[0] = gb[0];
  coef[7] = gb[(int64_t)4 * H];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float g = gb[(int64_t)(q + 1) * H];
    coef[1 + 2 * q] = g * (1.0f - r.w[q]);
    coef[2 + 2 * q] = g * r.w[q];
  }
  for (int t = 0; t < Tn; ++t) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < QN; ++j) v += (idx[j] == t) ? coef[j] : 0.f;
    db[(int64_t)t * H] = from_f32<T>(v);
  }
}

extern "C" int w2v2_pool_fwd(const void* x, float* out, int B, int T, int H, int mode, int dtype, void* stream) {
  W2V2_REQUIRE(x && out && B > 0 && T > 0 && H > 0 && mode >= 0 && (mode <= 5 || (mode >= 16 && mode - 16 < T)),
               "pool_fwd: bad arguments (mode %d, T %d)", mode, T);
  hipStream_t st = as_stream(stream);
  if (mode == 5) {
    dim3 grid((unsigned)cdiv((int64_t)B * H, 64));
    W2V2_DISPATCH_ACT(dtype, "pool_fwd",
      hipLaunchKernelGGL(pool_quantile_kernel<AT>, grid, dim3(64), 0, st, (const AT*)x, out, B, T, H););
  } else if (mode <= 1) {
    W2V2_REQUIRE(H % 8 == 0, "pool_fwd: H must be a multiple of 8");
    dim3 grid((unsigned)cdiv(H, 128), B);
    W2V2_DISPATCH_ACT(dtype, "pool_fwd",
      hipLaunchKernelGGL(pool_meanstd_kernel<AT>, grid, dim3(256), 0, st, (const AT*)x, out, T, H, mode == 0););
  } else {
    dim3 grid((unsigned)cdiv((int64_t)B * H, 256));
    W2V2_DISPATCH_ACT(dtype, "pool_fwd",
      hipLaunchKernelGGL(pool_select_kernel<AT>, grid, dim3(256), 0, st, (const AT*)x, out, B, T, H, mode););
  }
  W2V2_CHECK_LAUNCH("pool_fwd");
  return 0;
}

extern "C" int w2v2_pool_bwd(const void* x, const float* out, const float* dout, void* dx, int B, int T, int H,
                             int mode, int dtype, void* stream) {
  W2V2_REQUIRE(x && out && dout && dx && B > 0 && T > 0 && H > 0 && mode >= 0 &&
                   (mode <= 5 || (mode >= 16 && mode - 16 < T)), "pool_bwd: bad arguments (mode %d, T %d)", mode, T);
  hipStream_t st = as_stream(stream);
  if (mode == 5) {
    dim3 grid((unsigned)cdiv((int64_t)B * H, 64));
    W2V2_DISPATCH_ACT(dtype, "pool_bwd",
      hipLaunchKernelGGL(pool_quantile_bwd_kernel<AT>, grid, dim3(64), 0, st, (const AT*)x, dout, (AT*)dx, B, T, H););
  } else if (mode <= 1) {
    W2V2_REQUIRE(H % 8 == 0, "pool_bwd: H must be a multiple of 8");
    const int64_t total = (int64_t)B * T * (H >> 3);
    int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
    W2V2_DISPATCH_ACT(dtype, "pool_bwd",
      hipLaunchKernelGGL(pool_meanstd_bwd_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)x, out, dout,
                         (AT*)dx, B, T, H, mode == 0););
  } else {
    dim3 grid((unsigned)cdiv((int64_t)B * H, 256));
    W2V2_DISPATCH_ACT(dtype, "pool_bwd",
      hipLaunchKernelGGL(pool_select_bwd_kernel<AT>, grid, dim3(256), 0, st, (const AT*)x, dout, (AT*)dx,
                         B, T, H, mode););
  }
  W2V2_CHECK_LAUNCH("pool_bwd");
  return 0;
}
