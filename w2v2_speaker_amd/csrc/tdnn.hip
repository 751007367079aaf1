// tdnn.hip -- the non-GEMM pieces of the ECAPA-TDNN path (BASELINE configs[4];
// ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85 -> speechbrain 0.5.x ECAPA_TDNN, restated in
// oracle/ecapa_oracle.py -- speechbrain is not part of the reference tree: parity unpinned).
//
//   TDNNBlock  = Conv1d("same", reflect padding, dilation) -> ReLU -> BatchNorm1d (batch statistics)
//   conv       = im2col_reflect (taps gathered with reflection, channels-last) + GEMM (gemm.hip) [k = 1: GEMM only]
//   SE block   = mean_t -> Linear -> ReLU -> Linear -> sigmoid -> per-(utterance, channel) gate
//   Res2Net    = channel slices of one [B*T, C] tensor (row-strided views) with cumulative adds
//
// Everything here is HBM-bound elementwise / reduction work over channels-last [B*T, C] activations; reductions
// over rows are two-stage with a fixed order (deterministic, no atomics).  Row-strided operands (ld*) let the
// Res2Net slices and the MFA concatenation live inside their parent tensors without copies.
#include "common.cuh"

constexpr int TD_ROWS = 64;       // rows per partial-sum block of the BatchNorm reductions
constexpr int TD_TL = 8;          // time lanes of the per-(utterance, channel) walks

// ------------------------------------------------------------------------------------------ BatchNorm (+ReLU)
template <typename T>
__global__ void bn_partial_kernel(const T* __restrict__ a, int64_t lda, float* __restrict__ partial, int M, int C,
                                  int relu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.y * TD_ROWS, m1 = min(M, m0 + TD_ROWS);
  float s1 = 0.f, s2 = 0.f;
  for (int m = m0; m < m1; ++m) {
    float r = to_f32<T>(a[(int64_t)m * lda + c]);
    if (relu) r = fmaxf(r, 0.f);
    s1 += r;
    s2 = fmaf(r, r, s2);
  }
  float* pt = partial + ((int64_t)blockIdx.y * C + c) * 2;
  pt[0] = s1;
  pt[1] = s2;
}
// blockDim = (128 channels, 8 groups): group y folds partial blocks y, y+8, ... in order, thread y == 0 folds the groups
__device__ __forceinline__ void td_fold(const float* __restrict__ partial, int nblk, int C, int c, double& s1,
                                        double& s2, double (*red)[128][2]) {
  double p1 = 0.0, p2 = 0.0;
  if (c < C)
    for (int j = threadIdx.y; j < nblk; j += 8) {
      p1 += (double)partial[((int64_t)j * C + c) * 2];
      p2 += (double)partial[((int64_t)j * C + c) * 2 + 1];
    }
  red[threadIdx.y][threadIdx.x][0] = p1;
  red[threadIdx.y][threadIdx.x][1] = p2;
  __syncthreads();
  s1 = 0.0;
  s2 = 0.0;
  if (threadIdx.y == 0)
    for (int y = 0; y < 8; ++y) { s1 += red[y][threadIdx.x][0]; s2 += red[y][threadIdx.x][1]; }
}
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ mean_rstd, float* __restrict__ running,
                                                           int nblk, int M, int C, float eps, float momentum) {
  __shared__ double red[8][128][2];
  const int c = blockIdx.x * 128 + threadIdx.x;
  double s1, s2;
  td_fold(partial, nblk, C, c, s1, s2, red);
  if (threadIdx.y != 0 || c >= C) return;
  const double mu = s1 / M;
  double var = s2 / M - mu * mu;
  var = var > 0.0 ? var : 0.0;
  mean_rstd[2 * c] = (float)mu;
  mean_rstd[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (running != nullptr) {                       // torch BatchNorm1d buffers: momentum update, unbiased variance
    const double unb = M > 1 ? var * M / (M - 1) : var;
    running[c] = (1.f - momentum) * running[c] + momentum * (float)mu;
    running[C + c] = (1.f - momentum) * running[C + c] + momentum * (float)unb;
  }
}
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ a, int64_t lda, const float* __restrict__ mean_rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                int64_t ldy, int M, int C, int relu) {
  const int64_t n = (int64_t)M * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    float r = to_f32<T>(a[m * lda + c]);
    if (relu) r = fmaxf(r, 0.f);
    y[m * ldy + c] = from_f32<T>((r - mean_rstd[2 * c]) * mean_rstd[2 * c + 1] * gamma[c] + beta[c]);
  }
}
template <typename T>
__global__ void bn_bwd_partial_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ a, int64_t lda,
                                      const float* __restrict__ mean_rstd, float* __restrict__ partial, int M, int C,
                                      int relu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.y * TD_ROWS, m1 = min(M, m0 + TD_ROWS);
  const float mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int m = m0; m < m1; ++m) {
    float r = to_f32<T>(a[(int64_t)m * lda + c]);
    if (relu) r = fmaxf(r, 0.f);
    const float dz = to_f32<T>(dy[(int64_t)m * lddy + c]);
    s1 += dz;
    s2 = fmaf(dz, (r - mu) * rs, s2);
  }
  float* pt = partial + ((int64_t)blockIdx.y * C + c) * 2;
  pt[0] = s1;
  pt[1] = s2;
}
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int nblk, int C) {
  __shared__ double red[8][128][2];
  const int c = blockIdx.x * 128 + threadIdx.x;
  double s1, s2;
  td_fold(partial, nblk, C, c, s1, s2, red);
  if (threadIdx.y != 0 || c >= C) return;
  sums[2 * c] = (float)s1;
  sums[2 * c + 1] = (float)s2;
  dbeta[c] = (float)s1;                 // written, not accumulated
  dgamma[c] = (float)s2;
}
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy, const T* __restrict__ a, int64_t lda,
                                    const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums, T* __restrict__ da, int64_t ldda, int M, int C,
                                    int relu, float invM) {
  const int64_t n = (int64_t)M * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    const float av = to_f32<T>(a[m * lda + c]);
    const float r = relu ? fmaxf(av, 0.f) : av;
    const float rstd = mean_rstd[2 * c + 1];
    const float rh = (r - mean_rstd[2 * c]) * rstd;
    const float dz = to_f32<T>(dy[m * lddy + c]);
    const float dr = gamma[c] * rstd * (dz - sums[2 * c] * invM - rh * sums[2 * c + 1] * invM);
    da[m * ldda + c] = from_f32<T>((relu && !(av > 0.f)) ? 0.f : dr);
  }
}

// ------------------------------------------------------------------------------------------ reflect-padded im2col
// position t + (j - (k-1)/2) d, reflected at both ends (torch / speechbrain "reflect": ... 2 1 | 0 1 2 ... )
__device__ __forceinline__ int reflect(int p, int Tn) {
  if (p < 0) p = -p;
  if (p >= Tn) p = 2 * (Tn - 1) - p;
  return p;
}
// col[(b,t)][j*Cin + c] = x[b][reflect(t + off_j)][c];   8 channels per thread
template <typename T>
__global__ void im2col_reflect_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ col, int B, int Tn,
                                      int Cin, int k, int dil) {
  const int nch = Cin >> 3;
  const int64_t total = (int64_t)B * Tn * k * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    int64_t r = i / nch;
    const int j = (int)(r % k);
    r /= k;
    const int t = (int)(r % Tn), b = (int)(r / Tn);
    const int src = reflect(t + (j - (k - 1) / 2) * dil, Tn);
    Vec8<T> v;
    v.load(x + ((int64_t)b * Tn + src) * ldx + ch * 8);
    v.store(col + ((int64_t)b * Tn + t) * ((int64_t)k * Cin) + (int64_t)j * Cin + ch * 8);
  }
}
// dx[b][s][c] (+)= sum over (t, j) with reflect(t + off_j) == s of dcol[(b,t)][j*Cin + c]   (gather: deterministic)
template <typename T>
__global__ void col2im_reflect_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int64_t lddx, int B, int Tn,
                                      int Cin, int k, int dil, int accumulate) {
  const int nch = Cin >> 3;
  const int64_t total = (int64_t)B * Tn * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int s = (int)(row % Tn), b = (int)(row / Tn);
    float acc[8] = {};
    for (int j = 0; j < k; ++j) {
      const int off = (j - (k - 1) / 2) * dil;
      // sources t with reflect(t + off) == s: the direct one and the two mirror images
      const int cand[3] = {s - off, -s - off, 2 * (Tn - 1) - s - off};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int t = cand[q];
        if (t < 0 || t >= Tn) continue;
        const int p = t + off;
        const bool hit = q == 0 ? (p >= 0 && p < Tn) : q == 1 ? (p < 0) : (p >= Tn);
        if (!hit || reflect(p, Tn) != s) continue;
        Vec8<T> v;
        v.load(dcol + ((int64_t)b * Tn + t) * ((int64_t)k * Cin) + (int64_t)j * Cin + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v.v[e];
      }
    }
    T* dst = dx + row * lddx + ch * 8;
    Vec8<T> o;
    if (accumulate) {
      o.load(dst);
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] += acc[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = acc[e];
    }
    o.store(dst);
  }
}

// ------------------------------------------------------------------------------------------ strided add
template <typename T>
__global__ void add_strided_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                                   T* __restrict__ y, int64_t ldy, int M, int C) {
  const int nch = C >> 3;
  const int64_t total = (int64_t)M * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t m = i / nch;
    Vec8<T> va, vb;
    va.load(a + m * lda + ch * 8);
    vb.load(b + m * ldb + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) va.v[e] += vb.v[e];
    va.store(y + m * ldy + ch * 8);
  }
}

// ------------------------------------------------------------------------------------------ squeeze-excitation gate
__device__ __forceinline__ float td_block_sum(float v, float (*red)[64]) {
  __syncthreads();
  red[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < TD_TL; ++y) s += red[y][threadIdx.x];
  return s;
}
// y[b,t,c] = x[b,t,c] * g[b,c]
template <typename T>
__global__ void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ g, T* __restrict__ y, int B, int Tn,
                                int C) {
  const int64_t n = (int64_t)B * Tn * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((int64_t)Tn * C));
    y[i] = from_f32<T>(to_f32<T>(x[i]) * g[(int64_t)b * C + c]);
  }
}
// dg[b,c] = sum_t dout[b,t,c] * x[b,t,c]        blockDim = (64 channels, TD_TL time lanes)
template <typename T>
__global__ void se_bwd_gate_kernel(const T* __restrict__ dout, const T* __restrict__ x, float* __restrict__ dg, int Tn,
                                   int C) {
  __shared__ float red[TD_TL][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  const bool ok = c < C;
  float s = 0.f;
  if (ok)
    for (int t = threadIdx.y; t < Tn; t += TD_TL) {
      const int64_t i = ((int64_t)b * Tn + t) * C + c;
      s = fmaf(to_f32<T>(dout[i]), to_f32<T>(x[i]), s);
    }
  s = td_block_sum(s, red);
  if (ok && threadIdx.y == 0) dg[(int64_t)b * C + c] = s;
}
// dx[b,t,c] = dout[b,t,c] * g[b,c] + ds[b,c] / T      (ds = gradient wrt the time mean that feeds the gate)
template <typename T>
__global__ void se_bwd_x_kernel(const T* __restrict__ dout, const float* __restrict__ g, const float* __restrict__ ds,
                                T* __restrict__ dx, int B, int Tn, int C) {
  const int64_t n = (int64_t)B * Tn * C;
  const float invT = 1.0f / (float)Tn;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((int64_t)Tn * C));
    const int64_t bc = (int64_t)b * C + c;
    dx[i] = from_f32<T>(fmaf(to_f32<T>(dout[i]), g[bc], ds[bc] * invT));
  }
}

// ------------------------------------------------------------------------------------------ small f32 activations
// mode 0: relu, 1: sigmoid.  bwd takes the forward OUTPUT y.
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = mode == 0 ? fmaxf(v, 0.f) : 1.0f / (1.0f + __expf(-v));
  }
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               int64_t n, int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float yv = y[i];
    dx[i] = mode == 0 ? (yv > 0.f ? dy[i] : 0.f) : dy[i] * yv * (1.0f - yv);
  }
}

// ------------------------------------------------------------------------------------------ C ABI
static int td_blocks(int64_t n) { return (int)(cdiv(n, 256) > 8192 ? 8192 : (cdiv(n, 256) < 1 ? 1 : cdiv(n, 256))); }

extern "C" int w2v2_bn_workspace_floats(int M, int C) { return (int)cdiv(M, TD_ROWS) * C * 2 + 2 * C; }

extern "C" int w2v2_bn_stats(const void* a, int64_t lda, float* workspace, float* mean_rstd, float* running, int M,
                             int C, float eps, float momentum, int relu, int dtype, void* stream) {
  W2V2_REQUIRE(a && workspace && mean_rstd && M > 0 && C > 0 && lda >= C, "bn_stats: bad arguments");
  const int nblk = (int)cdiv(M, TD_ROWS);
  dim3 grid((unsigned)cdiv(C, 128), nblk);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "bn_stats",
    hipLaunchKernelGGL(bn_partial_kernel<AT>, grid, dim3(128), 0, st, (const AT*)a, lda, workspace, M, C, relu););
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128, 8), 0, st, workspace, mean_rstd,
                     running, nblk, M, C, eps, momentum);
  W2V2_CHECK_LAUNCH("bn_stats");
  return 0;
}

extern "C" int w2v2_bn_apply(const void* a, int64_t lda, const float* mean_rstd, const float* gamma, const float* beta,
                             void* y, int64_t ldy, int M, int C, int relu, int dtype, void* stream) {
  W2V2_REQUIRE(a && mean_rstd && gamma && beta && y && M > 0 && C > 0, "bn_apply: bad arguments");
  const int nb = td_blocks((int64_t)M * C);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "bn_apply",
    hipLaunchKernelGGL(bn_apply_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)a, lda, mean_rstd, gamma,
                           beta, (AT*)y, ldy, M, C, relu););
  W2V2_CHECK_LAUNCH("bn_apply");
  return 0;
}

extern "C" int w2v2_bn_bwd(const void* dy, int64_t lddy, const void* a, int64_t lda, const float* mean_rstd,
                           const float* gamma, float* workspace, float* dgamma, float* dbeta, void* da, int64_t ldda,
                           int M, int C, int relu, int dtype, void* stream) {
  W2V2_REQUIRE(dy && a && mean_rstd && gamma && workspace && dgamma && dbeta && da && M > 0 && C > 0,
               "bn_bwd: bad arguments");
  const int nblk = (int)cdiv(M, TD_ROWS);
  float* sums = workspace + (int64_t)nblk * C * 2;
  dim3 grid((unsigned)cdiv(C, 128), nblk);
  const int nb = td_blocks((int64_t)M * C);
  hipStream_t st = as_stream(stream);
#define TD_BNB(T_)                                                                                                  \
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T_>, grid, dim3(128), 0, st, (const T_*)dy, lddy, (const T_*)a, lda,     \
                     mean_rstd, workspace, M, C, relu);                                                             \
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128, 8), 0, st, workspace, sums,    \
                     dgamma, dbeta, nblk, C);                                                                       \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T_>, dim3(nb), dim3(256), 0, st, (const T_*)dy, lddy, (const T_*)a, lda,   \
                     mean_rstd, gamma, sums, (T_*)da, ldda, M, C, relu, 1.0f / (float)M)
  W2V2_DISPATCH_ACT(dtype, "bn_bwd", TD_BNB(AT););
#undef TD_BNB
  W2V2_CHECK_LAUNCH("bn_bwd");
  return 0;
}

extern "C" int w2v2_im2col_reflect(const void* x, int64_t ldx, void* col, int B, int T, int Cin, int k, int dilation,
                                   int dtype, void* stream) {
  W2V2_REQUIRE(x && col && B > 0 && T > 0 && Cin % 8 == 0 && k % 2 == 1 && dilation >= 1 && ldx % 8 == 0 &&
                   dilation * (k - 1) / 2 < T,
               "im2col_reflect: bad arguments (odd k, Cin %% 8 == 0, padding < T)");
  const int nb = td_blocks((int64_t)B * T * k * (Cin >> 3));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "im2col_reflect",
    hipLaunchKernelGGL(im2col_reflect_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)x, ldx,
                           (AT*)col, B, T, Cin, k, dilation););
  W2V2_CHECK_LAUNCH("im2col_reflect");
  return 0;
}

extern "C" int w2v2_col2im_reflect(const void* dcol, void* dx, int64_t lddx, int B, int T, int Cin, int k, int dilation,
                                   int accumulate, int dtype, void* stream) {
  W2V2_REQUIRE(dcol && dx && B > 0 && T > 0 && Cin % 8 == 0 && k % 2 == 1 && dilation >= 1 && lddx % 8 == 0 &&
                   dilation * (k - 1) / 2 < T,
               "col2im_reflect: bad arguments");
  const int nb = td_blocks((int64_t)B * T * (Cin >> 3));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "col2im_reflect",
    hipLaunchKernelGGL(col2im_reflect_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)dcol, (AT*)dx,
                           lddx, B, T, Cin, k, dilation, accumulate););
  W2V2_CHECK_LAUNCH("col2im_reflect");
  return 0;
}

extern "C" int w2v2_add_strided(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int M,
                                int C, int dtype, void* stream) {
  W2V2_REQUIRE(a && b && y && M > 0 && C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldy % 8 == 0,
               "add_strided: bad arguments");
  const int nb = td_blocks((int64_t)M * (C >> 3));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "add_strided",
    hipLaunchKernelGGL(add_strided_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)a, lda,
                           (const AT*)b, ldb, (AT*)y, ldy, M, C););
  W2V2_CHECK_LAUNCH("add_strided");
  return 0;
}

extern "C" int w2v2_se_scale(const void* x, const float* g, void* y, int B, int T, int C, int dtype, void* stream) {
  W2V2_REQUIRE(x && g && y && B > 0 && T > 0 && C > 0, "se_scale: bad arguments");
  const int nb = td_blocks((int64_t)B * T * C);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "se_scale",
    hipLaunchKernelGGL(se_scale_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)x, g, (AT*)y, B, T, C););
  W2V2_CHECK_LAUNCH("se_scale");
  return 0;
}

extern "C" int w2v2_se_bwd_gate(const void* dout, const void* x, float* dg, int B, int T, int C, int dtype,
                                void* stream) {
  W2V2_REQUIRE(dout && x && dg && B > 0 && T > 0 && C > 0, "se_bwd_gate: bad arguments");
  dim3 grid((unsigned)cdiv(C, 64), B), blk(64, TD_TL);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "se_bwd_gate",
    hipLaunchKernelGGL(se_bwd_gate_kernel<AT>, grid, blk, 0, st, (const AT*)dout, (const AT*)x, dg, T, C););
  W2V2_CHECK_LAUNCH("se_bwd_gate");
  return 0;
}

extern "C" int w2v2_se_bwd_x(const void* dout, const float* g, const float* ds, void* dx, int B, int T, int C, int dtype,
                             void* stream) {
  W2V2_REQUIRE(dout && g && ds && dx && B > 0 && T > 0 && C > 0, "se_bwd_x: bad arguments");
  const int nb = td_blocks((int64_t)B * T * C);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "se_bwd_x",
    hipLaunchKernelGGL(se_bwd_x_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)dout, g, ds, (AT*)dx,
                           B, T, C););
  W2V2_CHECK_LAUNCH("se_bwd_x");
  return 0;
}

extern "C" int w2v2_act_fwd(const float* x, float* y, int64_t n, int mode, void* stream) {
  W2V2_REQUIRE(x && y && n > 0 && (mode == 0 || mode == 1), "act_fwd: bad arguments");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(td_blocks(n)), dim3(256), 0, as_stream(stream), x, y, n, mode);
  W2V2_CHECK_LAUNCH("act_fwd");
  return 0;
}

extern "C" int w2v2_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int mode, void* stream) {
  W2V2_REQUIRE(dy && y && dx && n > 0 && (mode == 0 || mode == 1), "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(td_blocks(n)), dim3(256), 0, as_stream(stream), dy, y, dx, n, mode);
  W2V2_CHECK_LAUNCH("act_bwd");
  return 0;
}
