// norm.hip -- LayerNorm(+residual +dropout) forward / backward (HF:429, HF:596-601, HF:691-692).
// One 64-lane wave per token row, 16-byte vector accesses, wave-shuffle reductions; statistics f32.
#include "common.h"
#include <stdlib.h>

constexpr int LN_MAXC = 2;  // vec8 chunks per lane: H <= 1024
// workgroups of the backward (= rows of the gamma/beta partial buffer): three 4-wave workgroups per CU -- the 16-bit
// kernel holds two packed rows (current + prefetched) in 151 VGPRs
constexpr int LN_BWD_BLOCKS = 768;

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ r,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int M, int H, float eps, float p, uint64_t seed, int rounded_sum,
                                                     int gelu_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const int nch = H >> 3;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float s[LN_MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int ci = 0; ci < LN_MAXC; ++ci) {
    const int ch = lane + 64 * ci;
    if (ch < nch) {
      const int64_t off = (int64_t)row * H + ch * 8;
      Vec8<T> vx;
      vx.load(x + off);
      if (r != nullptr) {
        Vec8<T> vr;
        vr.load(r + off);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float s0 = 1.f, s1 = 1.f;
          if (p > 0.f) drop_scale2(seed, (uint64_t)(off + e), p, inv_keep, s0, s1);   // off is a multiple of 8
          vx.v[e] += vr.v[e] * s0;
          vx.v[e + 1] += vr.v[e + 1] * s1;
        }
        vx.store(r + off);  // pre-norm sum saved for backward (rounded to the activation format)
        // The statistics and y come from the UNROUNDED f32 sum: one rounding fewer on the forward path per LayerNorm
        // (24 per encoder pass).  The backward rebuilds x-hat from the stored 16-bit sum with these statistics: the
        // stored sum carries a rounding of 2^-12 |s| in fp16 and 2^-9 |s| in bf16, so x-hat (= (s - mean) * rstd) is off
        // by up to 2^-12 resp. 2^-9 times |s| * rstd -- in fp16 below the rounding of the gradients it multiplies; in
        // bf16 the same size as the rounding every other bf16 activation of the backward already carries (the bf16
        // gradient bounds of tests/test_parity_gpu.py, 12 %, are measured WITH it).  W2V2_LN_ROUNDED_SUM=1 restores
        // the old behaviour (statistics and y from the rounded sum).
        if (rounded_sum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vx.v[e] = to_f32<T>(from_f32<T>(vx.v[e]));
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[ci][e] = vx.v[e]; sum += vx.v[e]; }
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int ci = 0; ci < LN_MAXC; ++ci)
    if (lane + 64 * ci < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = s[ci][e] - mean; sq += d * d; }
  const float var = wave_sum(sq) / (float)H;
  const float rstd = rsqrtf(var + eps);
  if (lane == 0 && mean_o != nullptr) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int ci = 0; ci < LN_MAXC; ++ci) {
    const int ch = lane + 64 * ci;
    if (ch < nch) {
      Vec8<float> g, b;
      g.load(gamma + ch * 8);
      b.load(beta + ch * 8);
      Vec8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = (s[ci][e] - mean) * rstd * g.v[e] + b.v[e];
      if (gelu_out) {                 // the layer-norm convolution layers (HF:275-299): conv -> LayerNorm -> GELU
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = gelu_f(o.v[e]);
      }
      o.store(y + (int64_t)row * H + ch * 8);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 4) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, T* __restrict__ ds,
                                                     T* __restrict__ d_r, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ partial,
                                                     int M, int H, float p, uint64_t seed) {
  __shared__ float red[4][LN_MAXC * 64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float ag[LN_MAXC][8] = {}, ab[LN_MAXC][8] = {};
  float gm[LN_MAXC][8];
#pragma unroll
  for (int ci = 0; ci < LN_MAXC; ++ci) {
    const int ch = lane + 64 * ci;
    if (ch < nch) {
      Vec8<float> g;
      g.load(gamma + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) gm[ci][e] = g.v[e];
    }
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    // 16-bit activations: the row of s is kept PACKED (4 instead of 8 VGPRs per chunk) and unpacked again in the
    // second pass -- at 128 VGPRs (four workgroups per CU) the float copy spilled 5 registers, and a kernel that
    // touches scratch at all pays ~8 us extra per dispatch on this stack (tools/probes/scratch_probe.hip)
    constexpr bool PACK = sizeof(T) == 2;
    uint4 sraw[PACK ? LN_MAXC : 1];
    float xh[PACK ? 1 : LN_MAXC][8], g[LN_MAXC][8];
    auto xhat = [&](int ci, int e) -> float {
      if constexpr (PACK) {
        const uint32_t w = e < 2 ? sraw[ci].x : e < 4 ? sraw[ci].y : e < 6 ? sraw[ci].z : sraw[ci].w;
        float lo, hi;
        unpack2<T>(w, lo, hi);
        return ((e & 1 ? hi : lo) - mu) * rs;
      } else {
        return xh[ci][e];
      }
    };
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci) {
      const int ch = lane + 64 * ci;
      if (ch < nch) {
        const int64_t off = (int64_t)row * H + ch * 8;
        Vec8<T> vdy, vs;
        vdy.load(dy + off);
        vs.load(s + off);
        if constexpr (PACK) sraw[ci] = *reinterpret_cast<const uint4*>(s + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = (vs.v[e] - mu) * rs;
          if constexpr (!PACK) xh[ci][e] = xv;
          g[ci][e] = vdy.v[e] * gm[ci][e];
          c1 += g[ci][e];
          c2 += g[ci][e] * xv;
          ag[ci][e] += vdy.v[e] * xv;
          ab[ci][e] += vdy.v[e];
        }
      }
    }
    c1 = wave_sum(c1) / (float)H;
    c2 = wave_sum(c2) / (float)H;
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci) {
      const int ch = lane + 64 * ci;
      if (ch < nch) {
        const int64_t off = (int64_t)row * H + ch * 8;
        Vec8<T> o, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = rs * (g[ci][e] - c1 - xhat(ci, e) * c2);
        if (d_r != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            float s0 = 1.f, s1 = 1.f;
            if (p > 0.f) drop_scale2(seed, (uint64_t)(off + e), p, inv_keep, s0, s1);
            o2.v[e] = o.v[e] * s0;
            o2.v[e + 1] = o.v[e + 1] * s1;
          }
        }
        o.store(ds + off);
        if (d_r != nullptr) o2.store(d_r + off);
      }
    }
  }
  if (dgamma == nullptr && partial == nullptr) return;
  // cross-wave reduction of the per-lane column partials, then one atomic per column per block
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        red[wave][(ci * 64 + lane) * 8 + e] = pass == 0 ? ag[ci][e] : ab[ci][e];
    __syncthreads();
    for (int i = threadIdx.x; i < LN_MAXC * 64 * 8; i += 256) {
      const int ci = i / 512, rem = i - ci * 512, ln = rem >> 3, e = rem & 7;
      const int col = (ln + 64 * ci) * 8 + e;
      if (col < H) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (partial != nullptr) partial[((int64_t)blockIdx.x * 2 + pass) * H + col] = v;   // folded by ln_bwd_finalize
        else unsafeAtomicAdd((pass == 0 ? dgamma : dbeta) + col, v);
      }
    }
  }
}

// 16-bit activations: same arithmetic and the same per-block partial layout as ln_bwd_kernel, with the NEXT row's
// dy / s (and its mean / rstd) requested before the current row's two wave reductions: a wave walks only 2-3 rows at
// M = 9834, so without the prefetch every row is one exposed HBM round trip.  Both rows stay PACKED (4 VGPRs per
// 16-byte chunk); dy * gamma is recomputed in the second pass instead of kept (16 VGPRs).
template <typename T>
__global__ __launch_bounds__(256, 3) void ln_bwd16_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, T* __restrict__ ds,
                                                       T* __restrict__ d_r, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ partial,
                                                       int M, int H, float p, uint64_t seed) {
  static_assert(sizeof(T) == 2, "16-bit activations only");
  __shared__ float red[4][LN_MAXC * 64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float ag[LN_MAXC][8] = {}, ab[LN_MAXC][8] = {};
  float gm[LN_MAXC][8];
  bool have[LN_MAXC];
#pragma unroll
  for (int ci = 0; ci < LN_MAXC; ++ci) {
    have[ci] = lane + 64 * ci < nch;
#pragma unroll
    for (int e = 0; e < 8; ++e) gm[ci][e] = 0.f;
    if (have[ci]) {
      Vec8<float> g;
      g.load(gamma + (lane + 64 * ci) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) gm[ci][e] = g.v[e];
    }
  }
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  uint4 rdy[LN_MAXC], rs_[LN_MAXC], ndy[LN_MAXC], ns[LN_MAXC];
  float mu = 0.f, rs = 0.f, nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int r, uint4 (&a)[LN_MAXC], uint4 (&b)[LN_MAXC], float& m_, float& r_) {
    m_ = mean[r];
    r_ = rstd[r];
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci) {
      a[ci] = uint4{0u, 0u, 0u, 0u};
      b[ci] = uint4{0u, 0u, 0u, 0u};
      if (have[ci]) {
        const int64_t off = (int64_t)r * H + (lane + 64 * ci) * 8;
        a[ci] = *reinterpret_cast<const uint4*>(dy + off);
        b[ci] = *reinterpret_cast<const uint4*>(s + off);
      }
    }
  };
  auto word = [](const uint4& v, int i) -> uint32_t { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; };
  if (row < M) fetch(row, rdy, rs_, mu, rs);
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    if (nrow < M) fetch(nrow, ndy, ns, nmu, nrs);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float d0, d1, x0, x1;
        unpack2<T>(word(rdy[ci], w), d0, d1);
        unpack2<T>(word(rs_[ci], w), x0, x1);
        x0 = (x0 - mu) * rs;
        x1 = (x1 - mu) * rs;
        const float g0 = d0 * gm[ci][2 * w], g1 = d1 * gm[ci][2 * w + 1];
        c1 += g0;
        c2 += g0 * x0;
        c1 += g1;
        c2 += g1 * x1;
        ag[ci][2 * w] += d0 * x0;
        ag[ci][2 * w + 1] += d1 * x1;
        ab[ci][2 * w] += d0;
        ab[ci][2 * w + 1] += d1;
      }
    // (chunks a lane does not own hold dy = 0: they add nothing above; their x-hat is finite garbage times 0)
    c1 = wave_sum(c1) / (float)H;
    c2 = wave_sum(c2) / (float)H;
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci) {
      if (have[ci]) {
        const int64_t off = (int64_t)row * H + (lane + 64 * ci) * 8;
        uint32_t ow[4], ow2[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float d0, d1, x0, x1;
          unpack2<T>(word(rdy[ci], w), d0, d1);
          unpack2<T>(word(rs_[ci], w), x0, x1);
          x0 = (x0 - mu) * rs;
          x1 = (x1 - mu) * rs;
          const float o0 = rs * (d0 * gm[ci][2 * w] - c1 - x0 * c2);
          const float o1 = rs * (d1 * gm[ci][2 * w + 1] - c1 - x1 * c2);
          ow[w] = pack2<T>(o0, o1);
          if (d_r != nullptr) {
            float s0 = 1.f, s1 = 1.f;
            if (p > 0.f) drop_scale2(seed, (uint64_t)(off + 2 * w), p, inv_keep, s0, s1);
            ow2[w] = pack2<T>(o0 * s0, o1 * s1);
          }
        }
        store16_wt(ds + off, uint4{ow[0], ow[1], ow[2], ow[3]});
        if (d_r != nullptr) store16_wt(d_r + off, uint4{ow2[0], ow2[1], ow2[2], ow2[3]});
      }
    }
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci) { rdy[ci] = ndy[ci]; rs_[ci] = ns[ci]; }
    mu = nmu;
    rs = nrs;
  }
  if (dgamma == nullptr && partial == nullptr) return;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < LN_MAXC; ++ci)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        red[wave][(ci * 64 + lane) * 8 + e] = pass == 0 ? ag[ci][e] : ab[ci][e];
    __syncthreads();
    for (int i = threadIdx.x; i < LN_MAXC * 64 * 8; i += 256) {
      const int ci = i / 512, rem = i - ci * 512, ln = rem >> 3, e = rem & 7;
      const int col = (ln + 64 * ci) * 8 + e;
      if (col < H) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (partial != nullptr) partial[((int64_t)blockIdx.x * 2 + pass) * H + col] = v;   // folded by ln_bwd_finalize
        else unsafeAtomicAdd((pass == 0 ? dgamma : dbeta) + col, v);
      }
    }
  }
}

// The same for H = 256 NC (768: NC = 3, 512: NC = 2): a lane owns NC chunks of FOUR columns (8-byte accesses, lane l of a
// wave takes columns 256 c + 4 l ..).  With 16-byte chunks a 768-wide row is 96 chunks on 64 lanes: half of the wave idles
// through the second chunk and still holds its registers (151 VGPRs, three waves per SIMD).  Here every lane does the same
// work, the column accumulators shrink from 2 x 16 to 2 x 4 NC and the kernel runs at four waves per SIMD.
constexpr int LN_BWD_BLOCKS_Q = 1024;    // four workgroups per CU
template <typename T, int NC>
__global__ __launch_bounds__(256, 4) void ln_bwd16q_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, T* __restrict__ ds,
                                                        T* __restrict__ d_r, float* __restrict__ partial, int M,
                                                        float p, uint64_t seed) {
  static_assert(sizeof(T) == 2, "16-bit activations only");
  constexpr int H = 256 * NC;
  __shared__ float red[4][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  float ag[NC][4] = {}, ab[NC][4] = {}, gm[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + 256 * c + 4 * lane);
    gm[c][0] = g.x; gm[c][1] = g.y; gm[c][2] = g.z; gm[c][3] = g.w;
  }
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  uint2 rdy[NC], rs_[NC], ndy[NC], ns[NC];
  float mu = 0.f, rs = 0.f, nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int r, uint2 (&a)[NC], uint2 (&b)[NC], float& m_, float& r_) {
    m_ = mean[r];
    r_ = rstd[r];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t off = (int64_t)r * H + 256 * c + 4 * lane;
      a[c] = *reinterpret_cast<const uint2*>(dy + off);
      b[c] = *reinterpret_cast<const uint2*>(s + off);
    }
  };
  if (row < M) fetch(row, rdy, rs_, mu, rs);
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    if (nrow < M) fetch(nrow, ndy, ns, nmu, nrs);
    float c1 = 0.f, c2 = 0.f;
    float d[NC][4], xh[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      unpack2<T>(rdy[c].x, d[c][0], d[c][1]);
      unpack2<T>(rdy[c].y, d[c][2], d[c][3]);
      unpack2<T>(rs_[c].x, xh[c][0], xh[c][1]);
      unpack2<T>(rs_[c].y, xh[c][2], xh[c][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[c][e] = (xh[c][e] - mu) * rs;
        const float g = d[c][e] * gm[c][e];
        c1 += g;
        c2 = fmaf(g, xh[c][e], c2);
        ag[c][e] = fmaf(d[c][e], xh[c][e], ag[c][e]);
        ab[c][e] += d[c][e];
      }
    }
    c1 = wave_sum(c1) * (1.0f / (float)H);
    c2 = wave_sum(c2) * (1.0f / (float)H);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int64_t off = (int64_t)row * H + 256 * c + 4 * lane;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (d[c][e] * gm[c][e] - c1 - xh[c][e] * c2);
      *reinterpret_cast<uint2*>(ds + off) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
      if (d_r != nullptr) {
        float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f;
        if (p > 0.f) {
          drop_scale2(seed, (uint64_t)off, p, inv_keep, s0, s1);
          drop_scale2(seed, (uint64_t)(off + 2), p, inv_keep, s2, s3);
        }
        *reinterpret_cast<uint2*>(d_r + off) = make_uint2(pack2<T>(o[0] * s0, o[1] * s1), pack2<T>(o[2] * s2, o[3] * s3));
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) { rdy[c] = ndy[c]; rs_[c] = ns[c]; }
    mu = nmu;
    rs = nrs;
  }
  if (partial == nullptr) return;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][256 * c + 4 * lane + e] = pass == 0 ? ag[c][e] : ab[c][e];
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256)
      partial[((int64_t)blockIdx.x * 2 + pass) * H + col] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
  }
}

// fold the per-block column partials in a fixed order (deterministic) and add them to dgamma / dbeta:
// 32 columns x 32 block lanes per workgroup (round 5: 1024 threads; with 8 lanes a thread walked 96 partial rows and a
// four-LayerNorm fold took 11.6 us), each lane sums every 32nd block (independent loads, unrolled), then a fixed-order
// LDS fold over the 32 lanes
constexpr int LN_FOLD_LANES = 32;
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int nblk, int H) {
  __shared__ float red[LN_FOLD_LANES][32];
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;                 // index into [2][H]
  float s = 0.f;
  if (i < 2 * H) {
    const int pass = i / H, col = i - pass * H;
#pragma unroll 8
    for (int b = bl; b < nblk; b += LN_FOLD_LANES) s += partial[((int64_t)b * 2 + pass) * H + col];
  }
  red[bl][cl] = s;
  __syncthreads();
  if (bl == 0 && i < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_FOLD_LANES; ++k) t += red[k][cl];
    const int pass = i / H, col = i - pass * H;
    float* dst = (pass == 0 ? dgamma : dbeta) + col;
    *dst += t;
  }
}

// the same fold for up to 8 LayerNorms in one launch (blockIdx.y selects the entry): the engine defers the folds of a
// gradient bucket (two transformer blocks = four LayerNorms) to a single launch instead of four 6-us ones
struct LnFoldArgs {
  const float* partial[8];
  float* dgamma[8];
  float* dbeta[8];
};
__global__ __launch_bounds__(1024) void ln_bwd_finalize_many_kernel(const LnFoldArgs a, int nblk, int H) {
  __shared__ float red[LN_FOLD_LANES][32];
  const float* __restrict__ partial = a.partial[blockIdx.y];
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;                 // index into [2][H]
  float s = 0.f;
  if (i < 2 * H) {
    const int pass = i / H, col = i - pass * H;
#pragma unroll 8
    for (int b = bl; b < nblk; b += LN_FOLD_LANES) s += partial[((int64_t)b * 2 + pass) * H + col];
  }
  red[bl][cl] = s;
  __syncthreads();
  if (bl == 0 && i < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_FOLD_LANES; ++k) t += red[k][cl];
    const int pass = i / H, col = i - pass * H;
    float* dst = (pass == 0 ? a.dgamma[blockIdx.y] : a.dbeta[blockIdx.y]) + col;
    *dst += t;
  }
}

// workgroups of the backward (= partial blocks the fold walks): a function of (M, H) only, so that w2v2_layernorm_bwd_fold
// agrees with w2v2_layernorm_bwd whatever the dtype -- 1024 where the quad-chunk 16-bit kernel applies (four per CU)
static bool ln_quad_shape(int H) { return H == 512 || H == 768; }      // (NC = 4 spills at the 128-register cap: H = 1024 keeps the 16-byte kernel)
static int ln_bwd_nblocks(int M, int H) {
  const int cap = ln_quad_shape(H) ? LN_BWD_BLOCKS_Q : LN_BWD_BLOCKS;
  return (int)(cdiv(M, 4) < cap ? cdiv(M, 4) : cap);
}

extern "C" int w2v2_layernorm_bwd_fold(const w2v2_ln_fold* e, int n, int M, int H, void* stream) {
  W2V2_REQUIRE(e && n >= 0 && n <= 8 && H > 0, "layernorm_bwd_fold: bad arguments (at most 8 entries)");
  if (n == 0 || M <= 0) return 0;
  LnFoldArgs a;
  for (int i = 0; i < n; ++i) {
    W2V2_REQUIRE(e[i].partial && e[i].dgamma && e[i].dbeta, "layernorm_bwd_fold: null pointer in entry %d", i);
    a.partial[i] = e[i].partial; a.dgamma[i] = e[i].dgamma; a.dbeta[i] = e[i].dbeta;
  }
  const int nb = ln_bwd_nblocks(M, H);
  hipLaunchKernelGGL(ln_bwd_finalize_many_kernel, dim3((unsigned)cdiv(2 * H, 32), n), dim3(32 * LN_FOLD_LANES), 0, as_stream(stream),
                     a, nb, H);
  W2V2_CHECK_LAUNCH("layernorm_bwd_fold");
  return 0;
}

extern "C" int w2v2_layernorm_fwd(const void* x, void* r, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int M, int H, float eps, float drop_p,
                                  uint64_t seed, int dtype, void* stream) {
  W2V2_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  W2V2_REQUIRE(H % 8 == 0 && H <= 8 * 64 * LN_MAXC, "layernorm_fwd: H=%d unsupported (need H%%8==0, H<=1024)", H);
  W2V2_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_fwd: bad dropout p");
  if (M <= 0) return 0;
  static const bool rounded_sum = getenv("W2V2_LN_ROUNDED_SUM") != nullptr;      // A/B switch (parity experiments)
  dim3 grid((unsigned)cdiv(M, 4));
  W2V2_DISPATCH_ACT(dtype, "layernorm_fwd",
    hipLaunchKernelGGL(ln_fwd_kernel<AT>, grid, dim3(256), 0, as_stream(stream), (const AT*)x,
                       (AT*)r, gamma, beta, (AT*)y, mean, rstd, M, H, eps, drop_p, seed, (int)rounded_sum, 0););
  W2V2_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

extern "C" int w2v2_layernorm_gelu_fwd(const void* x, const float* gamma, const float* beta, void* y, int M, int H, float eps,
                                       int dtype, void* stream) {
  W2V2_REQUIRE(x && gamma && beta && y, "layernorm_gelu_fwd: null pointer");
  W2V2_REQUIRE(H % 8 == 0 && H <= 8 * 64 * LN_MAXC, "layernorm_gelu_fwd: H=%d unsupported (need H%%8==0, H<=1024)", H);
  if (M <= 0) return 0;
  dim3 grid((unsigned)cdiv(M, 4));
  W2V2_DISPATCH_ACT(dtype, "layernorm_gelu_fwd",
    hipLaunchKernelGGL(ln_fwd_kernel<AT>, grid, dim3(256), 0, as_stream(stream), (const AT*)x, (AT*)nullptr, gamma, beta,
                       (AT*)y, (float*)nullptr, (float*)nullptr, M, H, eps, 0.f, (uint64_t)0, 0, 1););
  W2V2_CHECK_LAUNCH("layernorm_gelu_fwd");
  return 0;
}

extern "C" int w2v2_layernorm_bwd_workspace_floats(int H) { return (LN_BWD_BLOCKS_Q > LN_BWD_BLOCKS ? LN_BWD_BLOCKS_Q : LN_BWD_BLOCKS) * 2 * H; }

extern "C" int w2v2_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd,
                                  const float* gamma, void* ds, void* d_r, float* dgamma, float* dbeta,
                                  float* workspace, int M, int H, float drop_p, uint64_t seed, int dtype,
                                  void* stream) {
  W2V2_REQUIRE(dy && s && mean && rstd && gamma && ds, "layernorm_bwd: null pointer");
  W2V2_REQUIRE(H % 8 == 0 && H <= 8 * 64 * LN_MAXC, "layernorm_bwd: H=%d unsupported", H);
  W2V2_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must come together");
  if (M <= 0) return 0;
  const int nb = ln_bwd_nblocks(M, H);
  // dgamma == NULL with a workspace: leave the per-block partials in it for w2v2_layernorm_bwd_fold
  float* partial = workspace;
  static const bool no_prefetch = getenv("W2V2_NO_LN_PREFETCH") != nullptr;       // A/B switch
  static const bool no_quad = getenv("W2V2_LN_NO_QUAD") != nullptr;               // A/B switch
  if (dtype != W2V2_F32 && !no_prefetch && !no_quad && ln_quad_shape(H) && partial != nullptr) {
    // 16-bit, H = 768 / 1024, per-block partials: the quad-chunk kernel (every lane busy, four waves per SIMD)
    W2V2_DISPATCH_16(dtype, "layernorm_bwd",
      if (H == 768)
        hipLaunchKernelGGL((ln_bwd16q_kernel<AT, 3>), dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)dy, (const AT*)s,
                           mean, rstd, gamma, (AT*)ds, (AT*)d_r, partial, M, drop_p, seed);
      else
        hipLaunchKernelGGL((ln_bwd16q_kernel<AT, 2>), dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)dy, (const AT*)s,
                           mean, rstd, gamma, (AT*)ds, (AT*)d_r, partial, M, drop_p, seed););
  } else if (dtype == W2V2_F32 || no_prefetch) {
    W2V2_DISPATCH_ACT(dtype, "layernorm_bwd",
      hipLaunchKernelGGL(ln_bwd_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)dy,
                         (const AT*)s, mean, rstd, gamma, (AT*)ds, (AT*)d_r, dgamma, dbeta, partial, M, H,
                         drop_p, seed););
  } else {
    W2V2_DISPATCH_16(dtype, "layernorm_bwd",
      hipLaunchKernelGGL(ln_bwd16_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)dy,
                         (const AT*)s, mean, rstd, gamma, (AT*)ds, (AT*)d_r, dgamma, dbeta, partial, M, H,
                         drop_p, seed););
  }
  if (partial != nullptr && dgamma != nullptr)
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((unsigned)cdiv(2 * H, 32)), dim3(32 * LN_FOLD_LANES), 0, as_stream(stream),
                       partial, dgamma, dbeta, nb, H);
  W2V2_CHECK_LAUNCH("layernorm_bwd");
  return 0;
}
