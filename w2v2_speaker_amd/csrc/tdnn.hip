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
#include "common.h"
#include <type_traits>

constexpr int BN_CW = 128;        // channels per workgroup of the BatchNorm kernels: 16 lanes x 8 channels (16 B)
constexpr int BN_RL = 16;         // row lanes of the 16-bit BatchNorm / SE workgroups (BnGeom below: 8 for f32)
constexpr int BN_AROWS = 256;     // rows one workgroup of the apply kernels walks (64 where that leaves CUs idle: bn_arows)
constexpr int BN_UN = 4;          // rows in flight per thread (memory-level parallelism of the row walks)

// ------------------------------------------------------------------------------------------ BatchNorm (+ReLU)
// Two launches per direction.  (1) partial: every workgroup reduces `rows` rows of a 128-channel strip to one
// (sum, sum of squares) pair per channel -- 16-byte loads, 16 row lanes, an LDS fold in a fixed order.  (2) apply:
// every workgroup first folds the partial blocks of ITS 128 channels (a few tens of KiB out of L2, in double, in
// block order -- the same numbers in every workgroup, so the result does not depend on the grid), then walks its
// rows.  The workgroups of row-block 0 publish mean / rstd (the backward reads them), update the running
// statistics, and in the backward write dgamma / dbeta.  No finalize launch, no atomics, no grid-wide fence.
constexpr int BN_PROWS = 256;     // rows per partial-sum block
__device__ __forceinline__ int bn_rows(int) { return BN_PROWS; }
static int bn_rows_host(int) { return BN_PROWS; }

// Thread geometry of the strip kernels.  A thread owns ONE 16-byte vector of a row: 8 channels of a 16-bit tensor, 4 of an
// f32 one -- round 6: with 8 f32 channels per thread (two 16-byte loads whose lanes sit 32 bytes apart) every load / store
// instruction used half of each 64-byte request and the f32 apply passes streamed 2.6 TB/s where a device copy of the
// same bytes streams 4.5 (profiles/r06_bn_rows_path.txt).  128 channels x RL row lanes = 256 threads either way.
struct Vec4f {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    store16_wt(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
  }
};
template <typename T> struct BnGeom {
  static constexpr int EPT = sizeof(T) == 4 ? 4 : 8;     // channels per thread
  static constexpr int XL = BN_CW / EPT;                 // channel lanes (blockDim.x)
  static constexpr int RL = 256 / XL;                    // row lanes (blockDim.y)
  static constexpr int UN = sizeof(T) == 4 ? 8 : 4;      // rows in flight per thread: 128 bytes either way
  using Vec = typename std::conditional<sizeof(T) == 4, Vec4f, Vec8<T>>::type;
  using FVec = typename std::conditional<sizeof(T) == 4, Vec4f, Vec8<float>>::type;   // f32 side operand, same channels
};
constexpr int BN_RL_MAX = 16;

// red[row lane][channel][2] -> partial[blockIdx.y][channel][2]
template <int EPT>
__device__ __forceinline__ void bn_store_partial(float (*red)[BN_CW][2], const float* s1, const float* s2,
                                                 float* __restrict__ partial, int C) {
  constexpr int XL = BN_CW / EPT, RL = 256 / XL;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    red[threadIdx.y][threadIdx.x * EPT + e][0] = s1[e];
    red[threadIdx.y][threadIdx.x * EPT + e][1] = s2[e];
  }
  __syncthreads();
  const int tid = threadIdx.y * XL + threadIdx.x;
  const int c = tid & (BN_CW - 1), w = tid >> 7;
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < RL; ++y) s += red[y][c][w];
  const int cc = blockIdx.x * BN_CW + c;
  if (cc < C) partial[((int64_t)blockIdx.y * C + cc) * 2 + w] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ a, int64_t lda,
                                                         float* __restrict__ partial, int M, int C, int relu) {
  __shared__ float red[BN_RL_MAX][BN_CW][2];
  using G = BnGeom<T>;
  constexpr int EPT = G::EPT, RL = G::RL, UN = G::UN;
  const int cg = blockIdx.x * BN_CW + threadIdx.x * EPT;
  const int rows = bn_rows(C);
  const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
  float s1[EPT] = {}, s2[EPT] = {};
  if (cg < C)
    for (int m = m0 + threadIdx.y; m < m1; m += UN * RL) {      // BN_UN independent 16-byte loads in flight
      typename G::Vec v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u].load(a + (int64_t)min(m + u * RL, m1 - 1) * lda + cg);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (m + u * RL < m1) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) {
            const float r = relu ? fmaxf(v[u].v[e], 0.f) : v[u].v[e];
            s1[e] += r;
            s2[e] = fmaf(r, r, s2[e]);
          }
        }
    }
  bn_store_partial<EPT>(red, s1, s2, partial, C);
}
// the two column sums of channel blockIdx.x*128 + (tid & 127), folded over the partial blocks in a fixed order: wave g
// takes blocks g, g+4, ... (one 16-byte load = 2 channels x 2 sums per lane, four loads in flight), then the four
// wave results are added in wave order
__device__ __forceinline__ void bn_fold(const float* __restrict__ partial, int nblk, int C, double (*fold)[BN_CW][2],
                                        double& s1, double& s2, int tid) {
  const int q = tid & 63, grp = tid >> 6;
  const int cq = blockIdx.x * BN_CW + 2 * q;
  double p[4] = {0.0, 0.0, 0.0, 0.0};
  if (cq < C) {
    const float* src = partial + (int64_t)cq * 2;
    int j = grp;
    for (; j + 12 < nblk; j += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(j + 4 * u) * C * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[0] += (double)v[u].x; p[1] += (double)v[u].y; p[2] += (double)v[u].z; p[3] += (double)v[u].w;
      }
    }
    for (; j < nblk; j += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)j * C * 2);
      p[0] += (double)v.x; p[1] += (double)v.y; p[2] += (double)v.z; p[3] += (double)v.w;
    }
  }
  fold[grp][2 * q][0] = p[0];
  fold[grp][2 * q][1] = p[1];
  fold[grp][2 * q + 1][0] = p[2];
  fold[grp][2 * q + 1][1] = p[3];
  __syncthreads();
  const int c = tid & (BN_CW - 1);
  s1 = ((fold[0][c][0] + fold[1][c][0]) + fold[2][c][0]) + fold[3][c][0];
  s2 = ((fold[0][c][1] + fold[1][c][1]) + fold[2][c][1]) + fold[3][c][1];
}
// mode 1: batch statistics from `partial` (training); mode 0: the running statistics (BatchNorm1d.eval())
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ partial, int nblk,
                                                       float* __restrict__ mean_rstd, float* __restrict__ running,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, int64_t ldy, int M, int C, int relu,
                                                       float eps, float momentum, int mode, int arows) {
  __shared__ double fold[4][BN_CW][2];
  __shared__ float cf[4][BN_CW];                 // mean, rstd, gamma, beta of the strip
  using G = BnGeom<T>;
  constexpr int EPT = G::EPT, RL = G::RL, UN = G::UN;
  const int tid = threadIdx.y * G::XL + threadIdx.x;
  const int c = tid & (BN_CW - 1), cc = blockIdx.x * BN_CW + c;
  float mu = 0.f, rstd = 0.f;
  if (mode) {
    double s1, s2;
    bn_fold(partial, nblk, C, fold, s1, s2, tid);
    const double mud = s1 / M;
    double var = s2 / M - mud * mud;
    var = var > 0.0 ? var : 0.0;
    mu = (float)mud;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.y == 0 && tid < BN_CW && cc < C && running != nullptr) {   // torch BatchNorm1d buffers: momentum
      const double unb = M > 1 ? var * M / (M - 1) : var;                  // update with the unbiased variance
      running[cc] = (1.f - momentum) * running[cc] + momentum * mu;
      running[C + cc] = (1.f - momentum) * running[C + cc] + momentum * (float)unb;
    }
  } else if (cc < C) {
    mu = running[cc];
    rstd = 1.0f / sqrtf(running[C + cc] + eps);
  }
  if (tid < BN_CW && cc < C) {
    cf[0][c] = mu;
    cf[1][c] = rstd;
    cf[2][c] = gamma[cc];
    cf[3][c] = beta[cc];
    if (blockIdx.y == 0) {
      mean_rstd[2 * cc] = mu;
      mean_rstd[2 * cc + 1] = rstd;
    }
  }
  __syncthreads();
  const int cl = threadIdx.x * EPT, cg = blockIdx.x * BN_CW + cl;
  if (cg >= C) return;
  const int m0 = blockIdx.y * arows, m1 = min(M, m0 + arows);
  float mus[EPT], rs[EPT], ga[EPT], be[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { mus[e] = cf[0][cl + e]; rs[e] = cf[1][cl + e]; ga[e] = cf[2][cl + e]; be[e] = cf[3][cl + e]; }
  for (int m = m0 + threadIdx.y; m < m1; m += UN * RL) {
    typename G::Vec v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u].load(a + (int64_t)min(m + u * RL, m1 - 1) * lda + cg);
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (m + u * RL < m1) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const float r = relu ? fmaxf(v[u].v[e], 0.f) : v[u].v[e];
          v[u].v[e] = (r - mus[e]) * rs[e] * ga[e] + be[e];
        }
        v[u].store(y + (int64_t)(m + u * RL) * ldy + cg);
      }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dy, int64_t lddy,
                                                             const T* __restrict__ dy2, int64_t lddy2,
                                                             const T* __restrict__ a, int64_t lda,
                                                             const float* __restrict__ mean_rstd,
                                                             float* __restrict__ partial, int M, int C, int relu) {
  __shared__ float red[BN_RL_MAX][BN_CW][2];
  using G = BnGeom<T>;
  constexpr int EPT = G::EPT, RL = G::RL, UN = G::UN;
  const int cg = blockIdx.x * BN_CW + threadIdx.x * EPT;
  const int rows = bn_rows(C);
  const int m0 = blockIdx.y * rows, m1 = min(M, m0 + rows);
  float s1[EPT] = {}, s2[EPT] = {};
  if (cg < C) {
    float mu[EPT], rs[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { mu[e] = mean_rstd[2 * (cg + e)]; rs[e] = mean_rstd[2 * (cg + e) + 1]; }
    for (int m = m0 + threadIdx.y; m < m1; m += UN * RL) {
      typename G::Vec v[UN], d[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t mm = min(m + u * RL, m1 - 1);
        v[u].load(a + mm * lda + cg);
        d[u].load(dy + mm * lddy + cg);
      }
      if (dy2 != nullptr) {                 // the gradient is dy + dy2 (two Res2Net branches), summed here
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          typename G::Vec w;
          w.load(dy2 + (int64_t)min(m + u * RL, m1 - 1) * lddy2 + cg);
#pragma unroll
          for (int e = 0; e < EPT; ++e) d[u].v[e] += w.v[e];
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (m + u * RL < m1) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) {
            const float r = relu ? fmaxf(v[u].v[e], 0.f) : v[u].v[e];
            s1[e] += d[u].v[e];
            s2[e] = fmaf(d[u].v[e], (r - mu[e]) * rs[e], s2[e]);
          }
        }
    }
  }
  bn_store_partial<EPT>(red, s1, s2, partial, C);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy,
                                                           const T* __restrict__ dy2, int64_t lddy2,
                                                           const T* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ partial, int nblk,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           T* __restrict__ da, int64_t ldda, int M, int C, int relu,
                                                           float invM, float* __restrict__ cs_partial, int arows) {
  __shared__ double fold[4][BN_CW][2];
  __shared__ float csred[BN_RL_MAX][BN_CW];          // column sums of da over this workgroup's rows (cs_partial != NULL)
  __shared__ float cf[5][BN_CW];                 // mean, rstd, gamma*rstd, sum dy / M, sum dy xhat / M
  using G = BnGeom<T>;
  constexpr int EPT = G::EPT, RL = G::RL, UN = G::UN;
  const int tid = threadIdx.y * G::XL + threadIdx.x;
  const int c = tid & (BN_CW - 1), cc = blockIdx.x * BN_CW + c;
  double s1, s2;
  bn_fold(partial, nblk, C, fold, s1, s2, tid);
  if (tid < BN_CW && cc < C) {
    const float f1 = (float)s1, f2 = (float)s2, rstd = mean_rstd[2 * cc + 1];
    cf[0][c] = mean_rstd[2 * cc];
    cf[1][c] = rstd;
    cf[2][c] = gamma[cc] * rstd;
    cf[3][c] = f1 * invM;
    cf[4][c] = f2 * invM;
    if (blockIdx.y == 0) {                       // written, not accumulated
      dbeta[cc] = f1;
      dgamma[cc] = f2;
    }
  }
  __syncthreads();
  const int cl = threadIdx.x * EPT, cg = blockIdx.x * BN_CW + cl;
  if (cg >= C && cs_partial == nullptr) return;
  const int m0 = blockIdx.y * arows, m1 = (cg < C) ? min(M, m0 + arows) : m0;
  float mu[EPT], rs[EPT], gr[EPT], m1s[EPT], m2s[EPT], cs[EPT] = {};
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    mu[e] = cf[0][cl + e]; rs[e] = cf[1][cl + e]; gr[e] = cf[2][cl + e]; m1s[e] = cf[3][cl + e]; m2s[e] = cf[4][cl + e];
  }
  for (int m = m0 + threadIdx.y; m < m1; m += UN * RL) {
    typename G::Vec v[UN], d[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t mm = min(m + u * RL, m1 - 1);
      v[u].load(a + mm * lda + cg);
      d[u].load(dy + mm * lddy + cg);
    }
    if (dy2 != nullptr) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        typename G::Vec w;
        w.load(dy2 + (int64_t)min(m + u * RL, m1 - 1) * lddy2 + cg);
#pragma unroll
        for (int e = 0; e < EPT; ++e) d[u].v[e] += w.v[e];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (m + u * RL < m1) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const float av = v[u].v[e];
          const float r = relu ? fmaxf(av, 0.f) : av;
          const float rh = (r - mu[e]) * rs[e];
          const float dr = gr[e] * (d[u].v[e] - m1s[e] - rh * m2s[e]);
          d[u].v[e] = (relu && !(av > 0.f)) ? 0.f : dr;
          cs[e] += d[u].v[e];
        }
        d[u].store(da + (int64_t)(m + u * RL) * ldda + cg);
      }
  }
  // column sums of da (= the bias gradient of the convolution in front of this BatchNorm) while the values are in
  // registers: per row-block partials, folded by the caller (w2v2_colsum over nblk rows instead of M)
  if (cs_partial != nullptr) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) csred[threadIdx.y][cl + e] = cs[e];
    __syncthreads();
    if (tid < BN_CW && cc < C) {
      float t = 0.f;
#pragma unroll
      for (int y = 0; y < RL; ++y) t += csred[y][c];
      cs_partial[(int64_t)blockIdx.y * C + cc] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------ reflect-padded im2col
// position t + (j - (k-1)/2) d, reflected at both ends (torch / speechbrain "reflect": ... 2 1 | 0 1 2 ... )
__device__ __forceinline__ int reflect(int p, int Tn) {
  if (p < 0) p = -p;
  if (p >= Tn) p = 2 * (Tn - 1) - p;
  return p;
}
// col[(b,t)][j*Cin + c] = x[b][reflect(t + off_j)][c];   one 16-byte vector per thread (8 / 4 channels)
template <typename T>
__global__ void im2col_reflect_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ x2, int64_t ldx2,
                                      T* __restrict__ col, int B, int Tn, int Cin, int k, int dil) {
  constexpr int EPT = BnGeom<T>::EPT;              // one 16-byte vector per thread (BnGeom)
  const int nch = Cin / EPT;
  const int64_t total = (int64_t)B * Tn * k * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    int64_t r = i / nch;
    const int j = (int)(r % k);
    r /= k;
    const int t = (int)(r % Tn), b = (int)(r / Tn);
    const int src = reflect(t + (j - (k - 1) / 2) * dil, Tn);
    typename BnGeom<T>::Vec v;
    v.load(x + ((int64_t)b * Tn + src) * ldx + ch * EPT);
    if (x2 != nullptr) {                   // taps of x + x2 (Res2Net: chunk input + previous chunk's output), summed here
      typename BnGeom<T>::Vec w;
      w.load(x2 + ((int64_t)b * Tn + src) * ldx2 + ch * EPT);
#pragma unroll
      for (int e = 0; e < EPT; ++e) v.v[e] += w.v[e];
    }
    v.store(col + ((int64_t)b * Tn + t) * ((int64_t)k * Cin) + (int64_t)j * Cin + ch * EPT);
  }
}
// dx[b][s][c] (+)= sum over (t, j) with reflect(t + off_j) == s of dcol[(b,t)][j*Cin + c]   (gather: deterministic)
template <typename T>
__global__ void col2im_reflect_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int64_t lddx, int B, int Tn,
                                      int Cin, int k, int dil, int accumulate) {
  constexpr int EPT = BnGeom<T>::EPT;              // one 16-byte vector per thread (BnGeom)
  const int nch = Cin / EPT;
  const int64_t total = (int64_t)B * Tn * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int s = (int)(row % Tn), b = (int)(row / Tn);
    float acc[EPT] = {};
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
        typename BnGeom<T>::Vec v;
        v.load(dcol + ((int64_t)b * Tn + t) * ((int64_t)k * Cin) + (int64_t)j * Cin + ch * EPT);
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[e] += v.v[e];
      }
    }
    T* dst = dx + row * lddx + ch * EPT;
    typename BnGeom<T>::Vec o;
    if (accumulate) {
      o.load(dst);
#pragma unroll
      for (int e = 0; e < EPT; ++e) o.v[e] += acc[e];
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) o.v[e] = acc[e];
    }
    o.store(dst);
  }
}

// ------------------------------------------------------------------------------------------ strided add
template <typename T>
__global__ void add_strided_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                                   T* __restrict__ y, int64_t ldy, int M, int C) {
  constexpr int EPT = BnGeom<T>::EPT;
  const int nch = C / EPT;
  const int64_t total = (int64_t)M * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t m = i / nch;
    typename BnGeom<T>::Vec va, vb;
    va.load(a + m * lda + ch * EPT);
    if (b != nullptr) {                    // b == NULL: a strided copy (Res2Net pass-through chunk, SE input slice)
      vb.load(b + m * ldb + ch * EPT);
#pragma unroll
      for (int e = 0; e < EPT; ++e) va.v[e] += vb.v[e];
    }
    va.store(y + m * ldy + ch * EPT);
  }
}

// ------------------------------------------------------------------------------------------ squeeze-excitation gate
// one 16-byte vector per thread (BnGeom: 8 channels of a 16-bit tensor, 4 of an f32 one); C % 8 == 0.
// y[b,t,c] = x[b,t,c] * g[b,c]
template <typename T>
__global__ void se_scale_kernel(const T* __restrict__ x, const float* __restrict__ g, T* __restrict__ y, int B, int Tn,
                                int C) {
  constexpr int EPT = BnGeom<T>::EPT;
  const int nch = C / EPT;
  const int total = B * Tn * nch;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / nch, ch = i - row * nch;
    const int b = row / Tn;
    typename BnGeom<T>::Vec v;
    typename BnGeom<T>::FVec gv;
    v.load(x + (int64_t)row * C + ch * EPT);
    gv.load(g + (int64_t)b * C + ch * EPT);
#pragma unroll
    for (int e = 0; e < EPT; ++e) v.v[e] *= gv.v[e];
    v.store(y + (int64_t)row * C + ch * EPT);
  }
}
// dg[b,c] = sum_t dout[b,t,c] * x[b,t,c]        blockDim = (BnGeom<T>::XL channel lanes, BnGeom<T>::RL time lanes)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_gate_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                          float* __restrict__ dg, int Tn, int C) {
  __shared__ float red[BN_RL_MAX][BN_CW];
  constexpr int EPT = BnGeom<T>::EPT, XL = BnGeom<T>::XL, RL = BnGeom<T>::RL;
  const int b = blockIdx.y, cg = blockIdx.x * BN_CW + threadIdx.x * EPT;
  float s[EPT] = {};
  if (cg < C)
    for (int t = threadIdx.y; t < Tn; t += RL) {
      const int64_t i = ((int64_t)b * Tn + t) * C + cg;
      typename BnGeom<T>::Vec d, v;
      d.load(dout + i);
      v.load(x + i);
#pragma unroll
      for (int e = 0; e < EPT; ++e) s[e] = fmaf(d.v[e], v.v[e], s[e]);
    }
#pragma unroll
  for (int e = 0; e < EPT; ++e) red[threadIdx.y][threadIdx.x * EPT + e] = s[e];
  __syncthreads();
  const int tid = threadIdx.y * XL + threadIdx.x;
  if (tid < BN_CW && blockIdx.x * BN_CW + tid < C) {
    float acc = 0.f;
#pragma unroll
    for (int yy = 0; yy < RL; ++yy) acc += red[yy][tid];
    dg[(int64_t)b * C + blockIdx.x * BN_CW + tid] = acc;
  }
}
// dx[b,t,c] = dout[b,t,c] * g[b,c] + ds[b,c] / T      (ds = gradient wrt the time mean that feeds the gate)
template <typename T>
__global__ void se_bwd_x_kernel(const T* __restrict__ dout, const float* __restrict__ g, const float* __restrict__ ds,
                                T* __restrict__ dx, int B, int Tn, int C) {
  constexpr int EPT = BnGeom<T>::EPT;
  const int nch = C / EPT;
  const int total = B * Tn * nch;
  const float invT = 1.0f / (float)Tn;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / nch, ch = i - row * nch;
    const int b = row / Tn;
    typename BnGeom<T>::Vec v;
    typename BnGeom<T>::FVec gv, sv;
    v.load(dout + (int64_t)row * C + ch * EPT);
    gv.load(g + (int64_t)b * C + ch * EPT);
    sv.load(ds + (int64_t)b * C + ch * EPT);
#pragma unroll
    for (int e = 0; e < EPT; ++e) v.v[e] = fmaf(v.v[e], gv.v[e], sv.v[e] * invT);
    v.store(dx + (int64_t)row * C + ch * EPT);
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
// (no cap on the grid: 8192 blocks made every thread of the 60 M-element passes walk 7 strides -- the Adam kernel lost 20 % to the same cap)
static int td_blocks(int64_t n) { return (int)(cdiv(n, 256) > (1 << 20) ? (1 << 20) : (cdiv(n, 256) < 1 ? 1 : cdiv(n, 256))); }

extern "C" int w2v2_bn_workspace_floats(int M, int C) { return (int)cdiv(M, 128) * C * 2; }
// Rows per workgroup of the apply kernels: 256, or 64 where 256 would leave most CUs without a workgroup -- the 128-channel
// Res2Net slices of the ECAPA path are ONE strip wide: 78 workgroups of 256 rows at M = 19800 ran a 10 MB pass in 17-19 us
// (four dependent load rounds behind the statistics fold on 78 of 256 CUs), 310 workgroups of 64 rows do one round each.
static int bn_arows(int M, int C) { return cdiv(C, BN_CW) * cdiv(M, BN_AROWS) >= 256 ? BN_AROWS : 64; }
extern "C" int w2v2_bn_colsum_rows(int M, int C) { return (int)cdiv(M, bn_arows(M, C)); }

extern "C" int w2v2_bn_fwd(const void* a, int64_t lda, float* workspace, float* mean_rstd, float* running,
                           const float* gamma, const float* beta, void* y, int64_t ldy, int M, int C, float eps,
                           float momentum, int relu, int train, int dtype, void* stream) {
  W2V2_REQUIRE(a && mean_rstd && gamma && beta && y && M > 0 && C > 0 && C % 8 == 0 && lda >= C && lda % 8 == 0 &&
                   ldy % 8 == 0, "bn_fwd: bad arguments (C, lda, ldy multiples of 8)");
  W2V2_REQUIRE(train ? workspace != nullptr : running != nullptr,
               "bn_fwd: training needs the workspace, evaluation the running statistics");
  const int nblk = (int)cdiv(M, bn_rows_host(C));
  const int arows = bn_arows(M, C);
  const dim3 gp((unsigned)cdiv(C, BN_CW), nblk), ga((unsigned)cdiv(C, BN_CW), (unsigned)cdiv(M, arows));
  hipStream_t st = as_stream(stream);
#define TD_BNF(T_)                                                                                                   \
  if (train)                                                                                                         \
    hipLaunchKernelGGL(bn_partial_kernel<T_>, gp, dim3(BnGeom<T_>::XL, BnGeom<T_>::RL), 0, st, (const T_*)a, lda, workspace, M, C, relu);             \
  hipLaunchKernelGGL(bn_apply_kernel<T_>, ga, dim3(BnGeom<T_>::XL, BnGeom<T_>::RL), 0, st, (const T_*)a, lda, workspace, nblk, mean_rstd, running,    \
                     gamma, beta, (T_*)y, ldy, M, C, relu, eps, momentum, train, arows)
  W2V2_DISPATCH_ACT(dtype, "bn_fwd", TD_BNF(AT););
#undef TD_BNF
  W2V2_CHECK_LAUNCH("bn_fwd");
  return 0;
}

static int bn_bwd_impl(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* a, int64_t lda,
                       const float* mean_rstd, const float* gamma, float* workspace, float* dgamma, float* dbeta, void* da,
                       int64_t ldda, int M, int C, int relu, float* colsum_partial, int dtype, void* stream) {
  W2V2_REQUIRE(dy && a && mean_rstd && gamma && workspace && dgamma && dbeta && da && M > 0 && C > 0 && C % 8 == 0 &&
                   lda % 8 == 0 && lddy % 8 == 0 && ldda % 8 == 0 && (dy2 == nullptr || lddy2 % 8 == 0),
               "bn_bwd: bad arguments (C and strides multiples of 8)");
  const int nblk = (int)cdiv(M, bn_rows_host(C));
  const int arows = bn_arows(M, C);
  const dim3 gp((unsigned)cdiv(C, BN_CW), nblk), ga((unsigned)cdiv(C, BN_CW), (unsigned)cdiv(M, arows));
  hipStream_t st = as_stream(stream);
#define TD_BNB(T_)                                                                                                  \
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T_>, gp, dim3(BnGeom<T_>::XL, BnGeom<T_>::RL), 0, st, (const T_*)dy, lddy, (const T_*)dy2, lddy2,         \
                     (const T_*)a, lda, mean_rstd, workspace, M, C, relu);                                          \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T_>, ga, dim3(BnGeom<T_>::XL, BnGeom<T_>::RL), 0, st, (const T_*)dy, lddy, (const T_*)dy2, lddy2,           \
                     (const T_*)a, lda, mean_rstd, gamma, workspace, nblk, dgamma, dbeta, (T_*)da, ldda, M, C, relu, \
                     1.0f / (float)M, colsum_partial, arows)
  W2V2_DISPATCH_ACT(dtype, "bn_bwd", TD_BNB(AT););
#undef TD_BNB
  W2V2_CHECK_LAUNCH("bn_bwd");
  return 0;
}
extern "C" int w2v2_bn_bwd(const void* dy, int64_t lddy, const void* a, int64_t lda, const float* mean_rstd,
                           const float* gamma, float* workspace, float* dgamma, float* dbeta, void* da, int64_t ldda,
                           int M, int C, int relu, float* colsum_partial, int dtype, void* stream) {
  return bn_bwd_impl(dy, lddy, nullptr, 0, a, lda, mean_rstd, gamma, workspace, dgamma, dbeta, da, ldda, M, C, relu,
                     colsum_partial, dtype, stream);
}
// the same with the output gradient given as dy + dy2 (two row-strided operands of one shape; both kernels add them as
// they read): the Res2Net chunk whose output feeds the next chunk AND the block's concatenation
extern "C" int w2v2_bn_bwd_sum(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* a, int64_t lda,
                               const float* mean_rstd, const float* gamma, float* workspace, float* dgamma, float* dbeta,
                               void* da, int64_t ldda, int M, int C, int relu, float* colsum_partial, int dtype,
                               void* stream) {
  W2V2_REQUIRE(dy2 != nullptr, "bn_bwd_sum: null second operand");
  return bn_bwd_impl(dy, lddy, dy2, lddy2, a, lda, mean_rstd, gamma, workspace, dgamma, dbeta, da, ldda, M, C, relu,
                     colsum_partial, dtype, stream);
}

static int im2col_reflect_impl(const void* x, int64_t ldx, const void* x2, int64_t ldx2, void* col, int B, int T, int Cin,
                               int k, int dilation, int dtype, void* stream) {
  W2V2_REQUIRE(x && col && B > 0 && T > 0 && Cin % 8 == 0 && k % 2 == 1 && dilation >= 1 && ldx % 8 == 0 &&
                   (x2 == nullptr || ldx2 % 8 == 0) && dilation * (k - 1) / 2 < T,
               "im2col_reflect: bad arguments (odd k, Cin %% 8 == 0, padding < T)");
  const int nb = td_blocks((int64_t)B * T * k * (Cin >> (dtype == W2V2_F32 ? 2 : 3)));   // one 16-byte vector per thread
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "im2col_reflect",
    hipLaunchKernelGGL(im2col_reflect_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)x, ldx, (const AT*)x2, ldx2,
                           (AT*)col, B, T, Cin, k, dilation););
  W2V2_CHECK_LAUNCH("im2col_reflect");
  return 0;
}
extern "C" int w2v2_im2col_reflect(const void* x, int64_t ldx, void* col, int B, int T, int Cin, int k, int dilation,
                                   int dtype, void* stream) {
  return im2col_reflect_impl(x, ldx, nullptr, 0, col, B, T, Cin, k, dilation, dtype, stream);
}
// the taps of x + x2 (same shape, own row strides): the Res2Net chunk input x_i + y_{i-1} without materialising the sum
extern "C" int w2v2_im2col_reflect_sum(const void* x, int64_t ldx, const void* x2, int64_t ldx2, void* col, int B, int T,
                                       int Cin, int k, int dilation, int dtype, void* stream) {
  W2V2_REQUIRE(x2 != nullptr, "im2col_reflect_sum: null second operand");
  return im2col_reflect_impl(x, ldx, x2, ldx2, col, B, T, Cin, k, dilation, dtype, stream);
}

extern "C" int w2v2_col2im_reflect(const void* dcol, void* dx, int64_t lddx, int B, int T, int Cin, int k, int dilation,
                                   int accumulate, int dtype, void* stream) {
  W2V2_REQUIRE(dcol && dx && B > 0 && T > 0 && Cin % 8 == 0 && k % 2 == 1 && dilation >= 1 && lddx % 8 == 0 &&
                   dilation * (k - 1) / 2 < T,
               "col2im_reflect: bad arguments");
  const int nb = td_blocks((int64_t)B * T * (Cin >> (dtype == W2V2_F32 ? 2 : 3)));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "col2im_reflect",
    hipLaunchKernelGGL(col2im_reflect_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)dcol, (AT*)dx,
                           lddx, B, T, Cin, k, dilation, accumulate););
  W2V2_CHECK_LAUNCH("col2im_reflect");
  return 0;
}

extern "C" int w2v2_add_strided(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int M,
                                int C, int dtype, void* stream) {
  W2V2_REQUIRE(a && y && M > 0 && C % 8 == 0 && lda % 8 == 0 && (b == nullptr || ldb % 8 == 0) && ldy % 8 == 0,
               "add_strided: bad arguments");
  const int nb = td_blocks((int64_t)M * (C >> (dtype == W2V2_F32 ? 2 : 3)));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "add_strided",
    hipLaunchKernelGGL(add_strided_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)a, lda,
                           (const AT*)b, ldb, (AT*)y, ldy, M, C););
  W2V2_CHECK_LAUNCH("add_strided");
  return 0;
}

extern "C" int w2v2_se_scale(const void* x, const float* g, void* y, int B, int T, int C, int dtype, void* stream) {
  W2V2_REQUIRE(x && g && y && B > 0 && T > 0 && C > 0 && C % 8 == 0, "se_scale: bad arguments (C %% 8 == 0)");
  const int nb = td_blocks((int64_t)B * T * (C >> (dtype == W2V2_F32 ? 2 : 3)));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "se_scale",
    hipLaunchKernelGGL(se_scale_kernel<AT>, dim3(nb), dim3(256), 0, st, (const AT*)x, g, (AT*)y, B, T, C););
  W2V2_CHECK_LAUNCH("se_scale");
  return 0;
}

extern "C" int w2v2_se_bwd_gate(const void* dout, const void* x, float* dg, int B, int T, int C, int dtype,
                                void* stream) {
  W2V2_REQUIRE(dout && x && dg && B > 0 && T > 0 && C > 0 && C % 8 == 0, "se_bwd_gate: bad arguments (C %% 8 == 0)");
  dim3 grid((unsigned)cdiv(C, BN_CW), B);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "se_bwd_gate",
    hipLaunchKernelGGL(se_bwd_gate_kernel<AT>, grid, dim3(BnGeom<AT>::XL, BnGeom<AT>::RL), 0, st, (const AT*)dout, (const AT*)x, dg, T, C););
  W2V2_CHECK_LAUNCH("se_bwd_gate");
  return 0;
}

extern "C" int w2v2_se_bwd_x(const void* dout, const float* g, const float* ds, void* dx, int B, int T, int C, int dtype,
                             void* stream) {
  W2V2_REQUIRE(dout && g && ds && dx && B > 0 && T > 0 && C > 0 && C % 8 == 0, "se_bwd_x: bad arguments (C %% 8 == 0)");
  const int nb = td_blocks((int64_t)B * T * (C >> (dtype == W2V2_F32 ? 2 : 3)));
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
