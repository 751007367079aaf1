// asp.hip -- attentive statistics pooling (ref call site: src/layers/pooling.py:87-106 ->
// speechbrain 0.5.x AttentiveStatisticsPooling(C, attention_channels=A=128, global_context=True);
// speechbrain is NOT under /root/reference: restated from the published definition, see oracle.attentive_stat_pool).
//
//   ctx   = [mean_t x, std_t x]                                  (biased var, clamp 1e-12)        [B, 2C]
//   a     = W1 . [x_t, ctx] + b1 = Wx x_t + (Wm mean + Ws std + b1)                               [B*T, A]
//   h     = tanh(BatchNorm_{batch stats over (B,T)}(relu(a)))                                     [B*T, A]
//   s     = W2 h + b2 ;  w = softmax_t(s)                                                         [B*T, C]
//   out   = [sum_t w x, sqrt(clamp(sum_t w (x - mean_w)^2, 1e-12))]   (mean FIRST, quirk Q1)      [B, 2C]
//
// The two matmuls (Wx x, W2 h) and their data / weight gradients run on the GEMM kernels (gemm.hip, wgrad.hip);
// this file holds the column / row statistics around them.  Everything here is HBM-bound and small (0.12 GFLOP/utt):
// one thread per (utterance, channel) walks the time axis, adjacent threads = adjacent channels (coalesced rows).
// Reductions over (B, T) are two-stage with a fixed order (deterministic, no atomics).
#include "common.h"

constexpr float ASP_EPS = 1e-12f;
// tanh through one exp + one rcp (|error| ~1e-7 relative to f32 libm tanhf, ~8 instead of ~40 instructions)
__device__ __forceinline__ float asp_tanh(float z) {
  const float a = fminf(fabsf(z), 15.0f);
  const float e = __expf(2.0f * a);
  return copysignf(1.0f - 2.0f * __frcp_rn(e + 1.0f), z);
}
constexpr int ASP_ROWS = 64;       // rows per partial-sum block of the BatchNorm reductions

// Per-(utterance, channel) walks over time use blockDim = (64 channels, ASP_TL time lanes): lane ty takes frames
// ty, ty + ASP_TL, ...; partial results meet in LDS and every thread folds them in the same fixed order.
constexpr int ASP_TL = 8;
__device__ __forceinline__ float asp_block_sum(float v, float (*red)[64]) {
  __syncthreads();
  red[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < ASP_TL; ++y) s += red[y][threadIdx.x];
  return s;
}
__device__ __forceinline__ float asp_block_max(float v, float (*red)[64]) {
  __syncthreads();
  red[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  float s = -INFINITY;
#pragma unroll
  for (int y = 0; y < ASP_TL; ++y) s = fmaxf(s, red[y][threadIdx.x]);
  return s;
}

// ------------------------------------------------------------------------------------------ global context
template <typename T>
__global__ void asp_context_kernel(const T* __restrict__ x, float* __restrict__ ctx, int Tn, int C) {
  __shared__ float red[ASP_TL][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  const bool ok = c < C;
  const T* xp = x + (int64_t)b * Tn * C + (ok ? c : 0);
  float s = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) s += to_f32<T>(xp[(int64_t)t * C]);
  const float mu = asp_block_sum(s, red) / (float)Tn;
  float q = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
      const float d = to_f32<T>(xp[(int64_t)t * C]) - mu;
      q = fmaf(d, d, q);
    }
  q = asp_block_sum(q, red);
  if (ok && threadIdx.y == 0) {
    ctx[(int64_t)b * 2 * C + c] = mu;
    ctx[(int64_t)b * 2 * C + C + c] = sqrtf(fmaxf(q / (float)Tn, ASP_EPS));
  }
}

// cb[b][a] = b1[a] + sum_j ctx[b][j] * W1[a][C + j]        (one wave per output)
__global__ __launch_bounds__(64) void asp_ctx_bias_kernel(const float* __restrict__ ctx, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, float* __restrict__ cb, int A,
                                                          int C) {
  const int a = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* cp = ctx + (int64_t)b * 2 * C;
  const float* wp = w1 + (int64_t)a * 3 * C + C;
  float s = 0.f;
  for (int j = lane; j < 2 * C; j += 64) s = fmaf(cp[j], wp[j], s);
  s = wave_sum(s);
  if (lane == 0) cb[(int64_t)b * A + a] = s + b1[a];
}

// ------------------------------------------------------------------------------------------ BatchNorm over (B,T)
// partial[blk][a][2] = {sum r, sum r^2} over ASP_ROWS rows, r = relu(a_pre)
template <typename T>
__global__ void asp_bn_partial_kernel(const T* __restrict__ a_pre, float* __restrict__ partial, int M, int A) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const int m0 = blockIdx.y * ASP_ROWS, m1 = min(M, m0 + ASP_ROWS);
  float s1 = 0.f, s2 = 0.f;
  for (int m = m0; m < m1; ++m) {
    const float r = fmaxf(to_f32<T>(a_pre[(int64_t)m * A + a]), 0.f);
    s1 += r;
    s2 = fmaf(r, r, s2);
  }
  float* pt = partial + ((int64_t)blockIdx.y * A + a) * 2;
  pt[0] = s1;
  pt[1] = s2;
}
// fixed-order fold -> {mean, rstd} (biased variance); running stats as torch BatchNorm1d (unbiased var, momentum).
// blockDim = (128 channels, 8 partial-block groups): group y folds blocks y, y+8, ... in order, then thread y == 0
// folds the 8 group sums in order (deterministic, 8x shorter serial chain than one thread per channel).
__device__ __forceinline__ void asp_fold(const float* __restrict__ partial, int nblk, int A, int a, double& s1,
                                         double& s2, double (*red)[128][2]) {
  double p1 = 0.0, p2 = 0.0;
  if (a < A)
    for (int j = threadIdx.y; j < nblk; j += 8) {
      p1 += (double)partial[((int64_t)j * A + a) * 2];
      p2 += (double)partial[((int64_t)j * A + a) * 2 + 1];
    }
  red[threadIdx.y][threadIdx.x][0] = p1;
  red[threadIdx.y][threadIdx.x][1] = p2;
  __syncthreads();
  s1 = 0.0;
  s2 = 0.0;
  if (threadIdx.y == 0)
    for (int y = 0; y < 8; ++y) { s1 += red[y][threadIdx.x][0]; s2 += red[y][threadIdx.x][1]; }
}
__global__ __launch_bounds__(1024) void asp_bn_finalize_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ mean_rstd,
                                                               float* __restrict__ running, int nblk, int M, int A,
                                                               float eps, float momentum) {
  __shared__ double red[8][128][2];
  const int a = blockIdx.x * 128 + threadIdx.x;
  double s1, s2;
  asp_fold(partial, nblk, A, a, s1, s2, red);
  if (threadIdx.y != 0 || a >= A) return;
  const double mu = s1 / M;
  double var = s2 / M - mu * mu;
  var = var > 0.0 ? var : 0.0;
  mean_rstd[2 * a] = (float)mu;
  mean_rstd[2 * a + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (running != nullptr) {
    const double unb = M > 1 ? var * M / (M - 1) : var;
    running[a] = (1.f - momentum) * running[a] + momentum * (float)mu;
    running[A + a] = (1.f - momentum) * running[A + a] + momentum * (float)unb;
  }
}
__global__ void asp_bn_eval_kernel(const float* __restrict__ running, float* __restrict__ mean_rstd, int A, float eps) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  mean_rstd[2 * a] = running[a];
  mean_rstd[2 * a + 1] = rsqrtf(running[A + a] + eps);
}

template <typename T>
__global__ void asp_bn_tanh_kernel(const T* __restrict__ a_pre, const float* __restrict__ mean_rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   T* __restrict__ h, int64_t n, int A) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const float r = fmaxf(to_f32<T>(a_pre[i]), 0.f);
    const float z = (r - mean_rstd[2 * a]) * mean_rstd[2 * a + 1] * gamma[a] + beta[a];
    h[i] = from_f32<T>(asp_tanh(z));
  }
}

// backward of h = tanh(BN(relu(a_pre))): partial sums of dz and dz * rhat, then the apply pass.  tanh(z) is
// RECOMPUTED from a_pre: 1 - h^2 from the stored bf16 h loses all precision for saturated units (h ~ 0.99 +- 0.004).
template <typename T>
__global__ void asp_bn_bwd_partial_kernel(const T* __restrict__ dh, const T* __restrict__ a_pre,
                                          const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ partial, int M, int A) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const int m0 = blockIdx.y * ASP_ROWS, m1 = min(M, m0 + ASP_ROWS);
  const float mu = mean_rstd[2 * a], rs = mean_rstd[2 * a + 1], ga = gamma[a], be = beta[a];
  float s1 = 0.f, s2 = 0.f;
  for (int m = m0; m < m1; ++m) {
    const int64_t i = (int64_t)m * A + a;
    const float rh = (fmaxf(to_f32<T>(a_pre[i]), 0.f) - mu) * rs;
    const float y = asp_tanh(fmaf(rh, ga, be));
    const float dz = to_f32<T>(dh[i]) * (1.f - y * y);
    s1 += dz;
    s2 = fmaf(dz, rh, s2);
  }
  float* pt = partial + ((int64_t)blockIdx.y * A + a) * 2;
  pt[0] = s1;
  pt[1] = s2;
}
// sums[a] = {sum dz, sum dz*rhat}; dbeta = sum dz, dgamma = sum dz*rhat (written, not accumulated)
__global__ __launch_bounds__(1024) void asp_bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ sums, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int nblk, int A) {
  __shared__ double red[8][128][2];
  const int a = blockIdx.x * 128 + threadIdx.x;
  double s1, s2;
  asp_fold(partial, nblk, A, a, s1, s2, red);
  if (threadIdx.y != 0 || a >= A) return;
  sums[2 * a] = (float)s1;
  sums[2 * a + 1] = (float)s2;
  dbeta[a] = (float)s1;
  dgamma[a] = (float)s2;
}
template <typename T>
__global__ void asp_bn_bwd_apply_kernel(const T* __restrict__ dh, const T* __restrict__ a_pre,
                                        const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, const float* __restrict__ sums,
                                        T* __restrict__ da, int64_t n, int A, float invM) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const float ap = to_f32<T>(a_pre[i]);
    const float mu = mean_rstd[2 * a], rstd = mean_rstd[2 * a + 1];
    const float rh = (fmaxf(ap, 0.f) - mu) * rstd;
    const float y = asp_tanh(fmaf(rh, gamma[a], beta[a]));
    const float dz = to_f32<T>(dh[i]) * (1.f - y * y);
    const float dr = gamma[a] * rstd * (dz - sums[2 * a] * invM - rh * sums[2 * a + 1] * invM);
    da[i] = from_f32<T>(ap > 0.f ? dr : 0.f);
  }
}

// ------------------------------------------------------------------------------------------ weighted statistics
// per (b, c): w = softmax_t(s);  out = [sum w x, sqrt(clamp(sum w (x - mean)^2))];  stats = {max_t s, sum_t exp(s - max)}
template <typename T>
__global__ void asp_pool_fwd_kernel(const T* __restrict__ x, const T* __restrict__ s, float* __restrict__ out,
                                    float* __restrict__ stats, int Tn, int C) {
  __shared__ float red[ASP_TL][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  const bool ok = c < C;
  const int64_t base = (int64_t)b * Tn * C + (ok ? c : 0);
  float mx = -INFINITY;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) mx = fmaxf(mx, to_f32<T>(s[base + (int64_t)t * C]));
  mx = asp_block_max(mx, red);
  float z = 0.f, m1 = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
      const float e = __expf(to_f32<T>(s[base + (int64_t)t * C]) - mx);
      z += e;
      m1 = fmaf(e, to_f32<T>(x[base + (int64_t)t * C]), m1);
    }
  z = asp_block_sum(z, red);
  m1 = asp_block_sum(m1, red);
  const float inv = 1.0f / z, mean = m1 * inv;
  float v = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
      const float e = __expf(to_f32<T>(s[base + (int64_t)t * C]) - mx);
      const float d = to_f32<T>(x[base + (int64_t)t * C]) - mean;
      v = fmaf(e * d, d, v);
    }
  v = asp_block_sum(v, red);
  if (ok && threadIdx.y == 0) {
    out[(int64_t)b * 2 * C + c] = mean;
    out[(int64_t)b * 2 * C + C + c] = sqrtf(fmaxf(v * inv, ASP_EPS));
    stats[((int64_t)b * C + c) * 2] = mx;
    stats[((int64_t)b * C + c) * 2 + 1] = z;
  }
}

// dout [B][2C] = {dmean, dstd}.  With var = sum w (x - mean)^2 (d var / d mean = 0 because sum w = 1):
//   dvar = dstd / (2 std) if var > eps;  dx_t = w_t (dmean + 2 dvar (x_t - mean));  dw_t = x_t dmean + dvar (x_t - mean)^2
//   ds_t = w_t (dw_t - sum_u w_u dw_u)
template <typename T>
__global__ void asp_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ s, const float* __restrict__ out,
                                    const float* __restrict__ stats, const float* __restrict__ dout,
                                    T* __restrict__ ds, T* __restrict__ dx, int Tn, int C) {
  __shared__ float red[ASP_TL][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  const bool ok = c < C;
  const int cc = ok ? c : 0;
  const int64_t base = (int64_t)b * Tn * C + cc;
  const float mean = out[(int64_t)b * 2 * C + cc], sd = out[(int64_t)b * 2 * C + C + cc];
  const float dmean = dout[(int64_t)b * 2 * C + cc], dstd = dout[(int64_t)b * 2 * C + C + cc];
  const float mx = stats[((int64_t)b * C + cc) * 2], inv = 1.0f / stats[((int64_t)b * C + cc) * 2 + 1];
  const float dvar = (sd * sd > ASP_EPS) ? dstd / (2.f * sd) : 0.f;
  float dot = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
      const float w = __expf(to_f32<T>(s[base + (int64_t)t * C]) - mx) * inv;
      const float xv = to_f32<T>(x[base + (int64_t)t * C]), d = xv - mean;
      dot = fmaf(w, xv * dmean + dvar * d * d, dot);
    }
  dot = asp_block_sum(dot, red);
  if (!ok) return;
  for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
    const float w = __expf(to_f32<T>(s[base + (int64_t)t * C]) - mx) * inv;
    const float xv = to_f32<T>(x[base + (int64_t)t * C]), d = xv - mean;
    const float dw = xv * dmean + dvar * d * d;
    ds[base + (int64_t)t * C] = from_f32<T>(w * (dw - dot));
    dx[base + (int64_t)t * C] = from_f32<T>(w * (dmean + 2.f * dvar * d));
  }
}

// ------------------------------------------------------------------------------------------ context backward
// dsum[b][a] = sum_t da[b,t,a]
template <typename T>
__global__ void asp_dsum_kernel(const T* __restrict__ da, float* __restrict__ dsum, int Tn, int A) {
  __shared__ float red[ASP_TL][64];
  const int b = blockIdx.y, a = blockIdx.x * 64 + threadIdx.x;
  const bool ok = a < A;
  float s = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += ASP_TL) s += to_f32<T>(da[((int64_t)b * Tn + t) * A + a]);
  s = asp_block_sum(s, red);
  if (ok && threadIdx.y == 0) dsum[(int64_t)b * A + a] = s;
}
// dW1[a][C + j] = sum_b dsum[b][a] ctx[b][j]  (j < 2C; written, not accumulated)
__global__ void asp_dw_ctx_kernel(const float* __restrict__ dsum, const float* __restrict__ ctx,
                                  float* __restrict__ dw1, int B, int A, int C) {
  const int a = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * C) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s = fmaf(dsum[(int64_t)b * A + a], ctx[(int64_t)b * 2 * C + j], s);
  dw1[(int64_t)a * 3 * C + C + j] = s;
}
// dctx[b][j] = sum_a dsum[b][a] W1[a][C + j]      blockDim = (64 columns j, 8 row groups of a), fixed-order fold
__global__ void asp_dctx_kernel(const float* __restrict__ dsum, const float* __restrict__ w1,
                                float* __restrict__ dctx, int A, int C) {
  __shared__ float red[ASP_TL][64];
  const int b = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  const bool ok = j < 2 * C;
  float s = 0.f;
  if (ok)
    for (int a = threadIdx.y; a < A; a += ASP_TL) s = fmaf(dsum[(int64_t)b * A + a], w1[(int64_t)a * 3 * C + C + j], s);
  s = asp_block_sum(s, red);
  if (ok && threadIdx.y == 0) dctx[(int64_t)b * 2 * C + j] = s;
}
// dx_t += dmean_ctx / T + dstd_ctx (x_t - mean) / (T std)    (std clamped: zero gradient below the clamp)
template <typename T>
__global__ void asp_context_bwd_kernel(const T* __restrict__ x, const float* __restrict__ ctx,
                                       const float* __restrict__ dctx, T* __restrict__ dx, int Tn, int C) {
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const int64_t base = (int64_t)b * Tn * C + c;
  const float mu = ctx[(int64_t)b * 2 * C + c], sd = ctx[(int64_t)b * 2 * C + C + c];
  const float dm = dctx[(int64_t)b * 2 * C + c] / (float)Tn;
  const float dsd = (sd * sd > ASP_EPS) ? dctx[(int64_t)b * 2 * C + C + c] / ((float)Tn * sd) : 0.f;
  for (int t = threadIdx.y; t < Tn; t += ASP_TL) {
    const int64_t i = base + (int64_t)t * C;
    const float xv = to_f32<T>(x[i]);
    dx[i] = from_f32<T>(to_f32<T>(dx[i]) + dm + dsd * (xv - mu));
  }
}

// ------------------------------------------------------------------------------------------ C ABI

extern "C" int w2v2_asp_context(const void* x, float* ctx, int B, int T, int C, int dtype, void* stream) {
  W2V2_REQUIRE(x && ctx && B > 0 && T > 0 && C > 0, "asp_context: bad arguments");
  dim3 grid((unsigned)cdiv(C, 64), B), blk(64, ASP_TL);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_context",
    hipLaunchKernelGGL(asp_context_kernel<AT>, grid, blk, 0, st, (const AT*)x, ctx, T, C););
  W2V2_CHECK_LAUNCH("asp_context");
  return 0;
}

extern "C" int w2v2_asp_context_bias(const float* ctx, const float* w1, const float* b1, float* cb, int B, int A,
                                     int C, void* stream) {
  W2V2_REQUIRE(ctx && w1 && b1 && cb && B > 0 && A > 0 && C > 0, "asp_context_bias: bad arguments");
  hipLaunchKernelGGL(asp_ctx_bias_kernel, dim3(A, B), dim3(64), 0, as_stream(stream), ctx, w1, b1, cb, A, C);
  W2V2_CHECK_LAUNCH("asp_context_bias");
  return 0;
}

extern "C" int w2v2_asp_bn_workspace_floats(int M, int A) { return (int)cdiv(M, ASP_ROWS) * A * 2 + 2 * A; }

extern "C" int w2v2_asp_bn_stats(const void* a_pre, float* workspace, float* mean_rstd, float* running, int M, int A,
                                 float eps, float momentum, int dtype, void* stream) {
  W2V2_REQUIRE(a_pre && workspace && mean_rstd && M > 0 && A > 0, "asp_bn_stats: bad arguments");
  const int nblk = (int)cdiv(M, ASP_ROWS);
  dim3 grid((unsigned)cdiv(A, 128), nblk);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_bn_stats",
    hipLaunchKernelGGL(asp_bn_partial_kernel<AT>, grid, dim3(128), 0, st, (const AT*)a_pre, workspace, M, A););
  hipLaunchKernelGGL(asp_bn_finalize_kernel, dim3((unsigned)cdiv(A, 128)), dim3(128, 8), 0, st, workspace, mean_rstd,
                     running, nblk, M, A, eps, momentum);
  W2V2_CHECK_LAUNCH("asp_bn_stats");
  return 0;
}

extern "C" int w2v2_asp_bn_eval_stats(const float* running, float* mean_rstd, int A, float eps, void* stream) {
  W2V2_REQUIRE(running && mean_rstd && A > 0, "asp_bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(asp_bn_eval_kernel, dim3((unsigned)cdiv(A, 128)), dim3(128), 0, as_stream(stream), running,
                     mean_rstd, A, eps);
  W2V2_CHECK_LAUNCH("asp_bn_eval_stats");
  return 0;
}

extern "C" int w2v2_asp_bn_tanh(const void* a_pre, const float* mean_rstd, const float* gamma, const float* beta,
                                void* h, int M, int A, int dtype, void* stream) {
  W2V2_REQUIRE(a_pre && mean_rstd && gamma && beta && h && M > 0 && A > 0, "asp_bn_tanh: bad arguments");
  const int64_t n = (int64_t)M * A;
  const int nb = (int)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_bn_tanh",
    hipLaunchKernelGGL(asp_bn_tanh_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)a_pre, mean_rstd,
                            gamma, beta, (AT*)h, n, A););
  W2V2_CHECK_LAUNCH("asp_bn_tanh");
  return 0;
}

extern "C" int w2v2_asp_bn_bwd(const void* dh, const void* a_pre, const float* mean_rstd, const float* gamma,
                               const float* beta, float* workspace, float* dgamma, float* dbeta, void* da, int M, int A,
                               int dtype, void* stream) {
  W2V2_REQUIRE(dh && a_pre && mean_rstd && gamma && beta && workspace && dgamma && dbeta && da && M > 0 && A > 0,
               "asp_bn_bwd: bad arguments");
  const int nblk = (int)cdiv(M, ASP_ROWS);
  float* sums = workspace + (int64_t)nblk * A * 2;
  dim3 grid((unsigned)cdiv(A, 128), nblk);
  const int64_t n = (int64_t)M * A;
  const int nb = (int)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256));
  hipStream_t st = as_stream(stream);
#define ASP_BNB(T_)                                                                                                  \
  hipLaunchKernelGGL(asp_bn_bwd_partial_kernel<T_>, grid, dim3(128), 0, st, (const T_*)dh, (const T_*)a_pre,         \
                     mean_rstd, gamma, beta, workspace, M, A);                                                       \
  hipLaunchKernelGGL(asp_bn_bwd_finalize_kernel, dim3((unsigned)cdiv(A, 128)), dim3(128, 8), 0, st, workspace, sums,    \
                     dgamma, dbeta, nblk, A);                                                                        \
  hipLaunchKernelGGL(asp_bn_bwd_apply_kernel<T_>, dim3(nb), dim3(256), 0, st, (const T_*)dh, (const T_*)a_pre,       \
                     mean_rstd, gamma, beta, sums, (T_*)da, n, A, 1.0f / (float)M)
  W2V2_DISPATCH_ACT(dtype, "asp_bn_bwd", ASP_BNB(AT););
#undef ASP_BNB
  W2V2_CHECK_LAUNCH("asp_bn_bwd");
  return 0;
}

extern "C" int w2v2_asp_pool_fwd(const void* x, const void* s, float* out, float* stats, int B, int T, int C, int dtype,
                                 void* stream) {
  W2V2_REQUIRE(x && s && out && stats && B > 0 && T > 0 && C > 0, "asp_pool_fwd: bad arguments");
  dim3 grid((unsigned)cdiv(C, 64), B), blk(64, ASP_TL);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_pool_fwd",
    hipLaunchKernelGGL(asp_pool_fwd_kernel<AT>, grid, blk, 0, st, (const AT*)x, (const AT*)s,
                            out, stats, T, C););
  W2V2_CHECK_LAUNCH("asp_pool_fwd");
  return 0;
}

extern "C" int w2v2_asp_pool_bwd(const void* x, const void* s, const float* out, const float* stats, const float* dout,
                                 void* ds, void* dx, int B, int T, int C, int dtype, void* stream) {
  W2V2_REQUIRE(x && s && out && stats && dout && ds && dx && B > 0 && T > 0 && C > 0, "asp_pool_bwd: bad arguments");
  dim3 grid((unsigned)cdiv(C, 64), B), blk(64, ASP_TL);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_pool_bwd",
    hipLaunchKernelGGL(asp_pool_bwd_kernel<AT>, grid, blk, 0, st, (const AT*)x, (const AT*)s,
                            out, stats, dout, (AT*)ds, (AT*)dx, T, C););
  W2V2_CHECK_LAUNCH("asp_pool_bwd");
  return 0;
}

// scratch: B*A (dsum) + B*2C (dctx) floats
extern "C" int w2v2_asp_context_bwd(const void* x, const float* ctx, const void* da, const float* w1, float* dw1,
                                    void* dx, float* scratch, int B, int T, int C, int A, int dtype, void* stream) {
  W2V2_REQUIRE(x && ctx && da && w1 && dw1 && dx && scratch && B > 0 && T > 0 && C > 0 && A > 0,
               "asp_context_bwd: bad arguments");
  float* dsum = scratch;
  float* dctx = scratch + (int64_t)B * A;
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "asp_context_bwd",
    hipLaunchKernelGGL(asp_dsum_kernel<AT>, dim3((unsigned)cdiv(A, 64), B), dim3(64, ASP_TL), 0, st,
                       (const AT*)da, dsum, T, A););
  hipLaunchKernelGGL(asp_dw_ctx_kernel, dim3((unsigned)cdiv(2 * C, 256), A), dim3(256), 0, st, dsum, ctx, dw1, B, A, C);
  hipLaunchKernelGGL(asp_dctx_kernel, dim3((unsigned)cdiv(2 * C, 64), B), dim3(64, ASP_TL), 0, st, dsum, w1, dctx, A, C);
  dim3 grid((unsigned)cdiv(C, 64), B), blk(64, ASP_TL);
  W2V2_DISPATCH_ACT(dtype, "asp_context_bwd",
    hipLaunchKernelGGL(asp_context_bwd_kernel<AT>, grid, blk, 0, st, (const AT*)x, ctx, dctx, (AT*)dx, T, C););
  W2V2_CHECK_LAUNCH("asp_context_bwd");
  return 0;
}
