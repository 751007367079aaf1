// pool.hip -- temporal pooling of the encoder output (ref: src/layers/pooling.py:24-44,74-80,118-136).
// x [B,T,H] channels-last -> f32 embedding.  mean+std: a workgroup owns (utterance, 128 channels);
// 16 column lanes x 16-byte vectors give fully coalesced 256-B row reads, the 16 time lanes
// (4 lane groups x 4 waves) are folded with wave shuffles (xor 16, 32) and one LDS hop across waves.
// Two passes over the (L2-resident) slab: mean, then centred sum of squares -> torch.std_mean parity.
#include "common.cuh"

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
  }
  for (int t = 0; t < Tn; ++t) db[(int64_t)t * H] = from_f32<T>(t == sel ? dout[i] : 0.f);
}

extern "C" int w2v2_pool_fwd(const void* x, float* out, int B, int T, int H, int mode, int dtype, void* stream) {
  W2V2_REQUIRE(x && out && B > 0 && T > 0 && H > 0 && mode >= 0 && mode <= 4, "pool_fwd: bad arguments");
  hipStream_t st = as_stream(stream);
  if (mode <= 1) {
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
  W2V2_REQUIRE(x && out && dout && dx && B > 0 && T > 0 && H > 0 && mode >= 0 && mode <= 4, "pool_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  if (mode <= 1) {
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
