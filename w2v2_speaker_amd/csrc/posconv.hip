// posconv.hip -- helpers around the grouped, weight-normed positional Conv1d (HF:326-379).
// The convolution itself runs on the MFMA GEMM as an implicit GEMM per group; these kernels build
// its operands: the padded group-major activation image and the weight-norm-ed packed weights, and
// the weight-norm backward.
#include "common.h"

// x [B,T,H] -> xg [B,G,Tp,Cg], Tp = T+K-1, xg[b,g,tp,c] = x[b, tp-pad_left, g*Cg+c] (0 outside).
// A row of the implicit GEMM for output frame t of group g is the contiguous K*Cg run starting at
// xg[b,g,t,0]:  lda = Cg, seg_len = T, seg_stride = G*Tp*Cg.
template <typename T>
__global__ void regroup_kernel(const T* __restrict__ x, T* __restrict__ xg, int B, int Tn, int H, int G, int K,
                               int pad_left) {
  const int Cg = H / G, Tp = Tn + K - 1, nch = Cg >> 3;
  const int64_t total = (int64_t)B * G * Tp * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    int64_t r = i / nch;
    const int tp = (int)(r % Tp); r /= Tp;
    const int g = (int)(r % G);
    const int b = (int)(r / G);
    const int t = tp - pad_left;
    Vec8<T> v;
    if (t >= 0 && t < Tn) {
      v.load(x + ((int64_t)b * Tn + t) * H + g * Cg + ch * 8);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v.v[e] = 0.f;
    }
    v.store(xg + (((int64_t)b * G + g) * Tp + tp) * Cg + ch * 8);
  }
}

extern "C" int w2v2_posconv_regroup(const void* x, void* xg, int B, int T, int H, int G, int K, int pad_left,
                                    int dtype, void* stream) {
  W2V2_REQUIRE(x && xg && B > 0 && T > 0 && G > 0 && H % G == 0 && (H / G) % 8 == 0 && K > 0,
               "posconv_regroup: bad arguments (H/G must be a multiple of 8)");
  const int64_t total = (int64_t)B * G * (T + K - 1) * ((H / G) >> 3);
  int nb = (int)(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256));
  W2V2_DISPATCH_ACT(dtype, "posconv_regroup",
    hipLaunchKernelGGL(regroup_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)x,
                       (AT*)xg, B, T, H, G, K, pad_left););
  W2V2_CHECK_LAUNCH("posconv_regroup");
  return 0;
}

// per-tap reductions over v [H][Cg][K] (K fastest): out[k] = sum_{o,i} a*b, in a FIXED order
// (per-block partials folded by a second kernel) so the packed weights are bitwise reproducible.
// Round 5: every kernel of this family walks its tensors along their CONTIGUOUS axis and does the layout change through
// an LDS tile -- the round-1 kernels wrote wf / wb as 2-byte elements 96 B apart and gathered dwf at a 9 KiB stride
// (wn_pack 48 us, tap_reduce<dW> 37 us, wn_bwd 28 us for 19 MB tensors); partial sums come from one block per output
// channel (H blocks instead of 128: the 144-trip dependent loops were latency-bound).
__host__ __device__ inline int wn_blocks(int H) { return H; }          // partial blocks = rows of the scratch past [0, K)

// sumsq partials: block o sums v[o][i][tap]^2 over i for every tap
__global__ __launch_bounds__(256) void tap_sumsq_kernel(const float* __restrict__ v, float* __restrict__ partial, int Cg,
                                                        int K) {
  __shared__ float red[256];
  const int rlanes = 256 / K, tap = threadIdx.x % K, rl = threadIdx.x / K;
  const float* vo = v + (int64_t)blockIdx.x * Cg * K;
  float acc = 0.f;
  if (rl < rlanes) {
    int i = rl;
    for (; i + 7 * rlanes < Cg; i += 8 * rlanes) {        // eight independent loads in flight; the plain loop's order
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = vo[(int64_t)(i + u * rlanes) * K + tap];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(x[u], x[u], acc);
    }
    for (; i < Cg; i += rlanes) { const float x = vo[(int64_t)i * K + tap]; acc = fmaf(x, x, acc); }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < K) {
    float s = 0.f;
    for (int j = 0; j < rlanes; ++j) s += red[j * K + threadIdx.x];
    partial[(int64_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// out[k] = sum_b partial[b][k] in a fixed order: one block per tap, 256 threads x ceil(nb / 256) partials, then the
// fixed-shape block fold (wave shuffles + 4 LDS slots)
__global__ __launch_bounds__(256) void tap_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int K,
                                                           int nb) {
  __shared__ float red[4];
  const int k = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) s += partial[(int64_t)b * K + k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Packed operands through an LDS tile.  blockIdx.y = 0: block o -> wf[o][tap][i] (the [Cg][K] slab v[o] is contiguous);
// blockIdx.y = 1: block (g, i) -> wb[(g, i)][K-1-tap][co] from the Cg rows v[(g, co)][i][:] (512-byte runs).
template <typename T>
__global__ __launch_bounds__(256) void wn_pack_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                      const float* __restrict__ sumsq, T* __restrict__ wf,
                                                      T* __restrict__ wb, int Cg, int K) {
  extern __shared__ float wn_tile[];                 // [Cg][K + 1], then the per-tap scale [K]
  const int blk = blockIdx.x, mode = blockIdx.y, P = K + 1;
  const int gi = blk / Cg, r = blk - gi * Cg;
  float* sc = wn_tile + Cg * P;
  for (int t = threadIdx.x; t < K; t += 256) sc[t] = g[t] * rsqrtf(sumsq[t]);
  // the slab in flight first (8 independent loads per thread and pass), scaled on the way out
  const int64_t row_stride = mode == 0 ? K : (int64_t)Cg * K;
  const float* src0 = v + (mode == 0 ? (int64_t)blk * Cg * K : ((int64_t)gi * Cg * Cg + r) * K);
  const int n = Cg * K;
  for (int e0 = threadIdx.x; e0 < n; e0 += 256 * 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      x[u] = e < n ? src0[(int64_t)(e / K) * row_stride + (e % K)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < n) wn_tile[(e / K) * P + (e % K)] = x[u];
    }
  }
  __syncthreads();
  T* dst = (mode == 0 ? wf : wb) + (int64_t)blk * K * Cg;
  if ((Cg & 7) == 0) {                               // 8 consecutive outputs share the tap row: one 16-byte store
    for (int e = threadIdx.x * 8; e < n; e += 256 * 8) {
      const int to = e / Cg, j = e - to * Cg, tap = mode == 0 ? to : K - 1 - to;
      const float s1 = sc[tap];
      Vec8<T> o;
#pragma unroll
      for (int u = 0; u < 8; ++u) o.v[u] = wn_tile[(j + u) * P + tap] * s1;
      o.store(dst + e);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 256) {
      const int to = e / Cg, j = e - to * Cg, tap = mode == 0 ? to : K - 1 - to;
      dst[e] = from_f32<T>(wn_tile[j * P + tap] * sc[tap]);
    }
  }
}

extern "C" int64_t w2v2_weightnorm_scratch_floats(int H, int G, int K) { (void)G; return (int64_t)(1 + wn_blocks(H)) * K; }

extern "C" int w2v2_weightnorm_pack(const float* g, const float* v, float* sumsq, void* wf, void* wb, int H, int G,
                                    int K, int dtype, void* stream) {
  W2V2_REQUIRE(g && v && sumsq && wf && wb && G > 0 && H % G == 0 && K > 0 && K <= 256, "weightnorm_pack: bad arguments");
  const int Cg = H / G;
  W2V2_REQUIRE(((size_t)Cg * (K + 1) + K) * 4 <= 160 * 1024, "weightnorm_pack: group tile does not fit LDS (Cg=%d K=%d)", Cg, K);
  hipStream_t st = as_stream(stream);
  const int nb = wn_blocks(H);
  hipLaunchKernelGGL(tap_sumsq_kernel, dim3(nb), dim3(256), 0, st, v, sumsq + K, Cg, K);
  hipLaunchKernelGGL(tap_finalize_kernel, dim3(K), dim3(256), 0, st, sumsq + K, sumsq, K, nb);
  const size_t lds = ((size_t)Cg * (K + 1) + K) * sizeof(float);
  W2V2_DISPATCH_ACT(dtype, "weightnorm_pack", {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_pack_kernel<AT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wn_pack_kernel<AT>, dim3(H, 2), dim3(256), lds, st, g, v, sumsq, (AT*)wf, (AT*)wb, Cg, K);
  });
  W2V2_CHECK_LAUNCH("weightnorm_pack");
  return 0;
}

// w = g_k v / n_k:  dg_k = dot_k / n_k,  dv = g_k/n_k * (dw - v * dot_k / n_k^2),  dot_k = sum dw*v
// Pass 1, block (g, i): the slab dwf[g][:][i][:] ([K] runs of Cg floats) goes through an LDS tile into dv in v's layout
// (dv[(g, co)][i][tap] = dw, 512-byte runs) and its contribution to dot[tap] = sum dw * v into the block's partial row.
__global__ __launch_bounds__(256) void wn_bwd_gather_kernel(const float* __restrict__ v, const float* __restrict__ dwf,
                                                            float* __restrict__ dv, float* __restrict__ partial, int Cg,
                                                            int K) {
  extern __shared__ float wn_tile[];                 // [K][Cg + 1], then the fold scratch [256]
  const int blk = blockIdx.x, P = Cg + 1;
  const int gi = blk / Cg, i = blk - gi * Cg;
  float* red = wn_tile + K * P;
#pragma unroll 8
  for (int e = threadIdx.x; e < K * Cg; e += 256) {
    const int tap = e / Cg, co = e - tap * Cg;
    wn_tile[tap * P + co] = dwf[(((int64_t)gi * K + tap) * Cg + i) * Cg + co];
  }
  __syncthreads();
  const int rlanes = 256 / K, tap = threadIdx.x % K, rl = threadIdx.x / K;
  float acc = 0.f;
  if (rl < rlanes)
    {
      int co = rl;
      for (; co + 7 * rlanes < Cg; co += 8 * rlanes) {    // eight v loads in flight
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = v[(((int64_t)gi * Cg + co + u * rlanes) * Cg + i) * K + tap];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float d = wn_tile[tap * P + co + u * rlanes];
          dv[(((int64_t)gi * Cg + co + u * rlanes) * Cg + i) * K + tap] = d;
          acc = fmaf(d, vv[u], acc);
        }
      }
      for (; co < Cg; co += rlanes) {
        const int64_t idx = (((int64_t)gi * Cg + co) * Cg + i) * K + tap;
        const float d = wn_tile[tap * P + co];
        dv[idx] = d;
        acc = fmaf(d, v[idx], acc);
      }
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < K) {
    float s = 0.f;
    for (int j = 0; j < rlanes; ++j) s += red[j * K + threadIdx.x];
    partial[(int64_t)blk * K + threadIdx.x] = s;
  }
}
// Pass 2, elementwise in v's layout (dv holds dw)
__global__ void wn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                    const float* __restrict__ sumsq, const float* __restrict__ dot,
                                    float* __restrict__ dg, float* __restrict__ dv, int64_t total, int K) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % K);
    const float n2 = sumsq[tap], rn = rsqrtf(n2);
    dv[idx] = g[tap] * rn * (dv[idx] - v[idx] * dot[tap] / n2);
    if (idx < K) dg[idx] = dot[idx] * rsqrtf(sumsq[idx]);
  }
}

extern "C" int w2v2_weightnorm_bwd(const float* g, const float* v, const float* sumsq, const float* dwf, float* dot,
                                   float* dg, float* dv, int H, int G, int K, void* stream) {
  W2V2_REQUIRE(g && v && sumsq && dwf && dot && dg && dv && G > 0 && H % G == 0 && K > 0 && K <= 256,
               "weightnorm_bwd: bad arguments");
  const int Cg = H / G;
  const size_t lds = ((size_t)K * (Cg + 1) + 256) * sizeof(float);
  W2V2_REQUIRE(lds <= 160 * 1024, "weightnorm_bwd: group tile does not fit LDS (Cg=%d K=%d)", Cg, K);
  hipStream_t st = as_stream(stream);
  const int nb = wn_blocks(H);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_bwd_gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(wn_bwd_gather_kernel, dim3(H), dim3(256), lds, st, v, dwf, dv, dot + K, Cg, K);
  hipLaunchKernelGGL(tap_finalize_kernel, dim3(K), dim3(256), 0, st, dot + K, dot, K, nb);
  const int64_t total = (int64_t)H * Cg * K;
  int nbk = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3(nbk), dim3(256), 0, st, g, v, sumsq, dot, dg, dv, total, K);
  W2V2_CHECK_LAUNCH("weightnorm_bwd");
  return 0;
}
