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
constexpr int TAP_BLOCKS = 128;

template <bool WITH_DW>
__global__ __launch_bounds__(256) void tap_reduce_kernel(const float* __restrict__ v, const float* __restrict__ dwf,
                                                         float* __restrict__ partial, int H, int Cg, int K) {
  __shared__ float red[256];
  // thread -> tap (k = tid % K); rows (o,i) strided over (block, row lane)
  const int rows = H * Cg;
  const int kpt = threadIdx.x % K, rlane = threadIdx.x / K, rlanes = 256 / K;
  float acc = 0.f;
  if (rlane < rlanes) {
    const int stride = gridDim.x * rlanes;
    int r = blockIdx.x * rlanes + rlane;
    // eight independent loads in flight per thread (the loop is latency-bound: 128 blocks x 144 dependent trips, and
    // the dwf operand is a 9 KiB-stride gather); the accumulation order is the plain loop's
    for (; r + 7 * stride < rows; r += 8 * stride) {
      float t[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ru = r + u * stride;
        t[u] = v[(int64_t)ru * K + kpt];
        if constexpr (WITH_DW) {
          const int o = ru / Cg, i = ru - o * Cg;
          const int g = o / Cg, co = o - g * Cg;
          d[u] = dwf[(((int64_t)g * K + kpt) * Cg + i) * Cg + co];
        } else {
          d[u] = t[u];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u] * d[u];
    }
    for (; r < rows; r += stride) {
      const float vv = v[(int64_t)r * K + kpt];
      if constexpr (WITH_DW) {
        const int o = r / Cg, i = r - o * Cg;
        const int g = o / Cg, co = o - g * Cg;
        acc += vv * dwf[(((int64_t)g * K + kpt) * Cg + i) * Cg + co];
      } else {
        acc += vv * vv;
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < K) {
    float s = 0.f;
    for (int j = 0; j < rlanes; ++j) s += red[j * K + threadIdx.x];
    partial[(int64_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

__global__ void tap_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int b = 0; b < TAP_BLOCKS; ++b) s += partial[(int64_t)b * K + k];
  out[k] = s;
}

template <typename T>
__global__ void wn_pack_kernel(const float* __restrict__ g, const float* __restrict__ v,
                               const float* __restrict__ sumsq, T* __restrict__ wf, T* __restrict__ wb, int H,
                               int Cg, int K) {
  const int64_t total = (int64_t)H * Cg * K;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % K);
    const int64_t r = idx / K;
    const int i = (int)(r % Cg), o = (int)(r / Cg);
    const int gi = o / Cg, co = o - gi * Cg;
    const float wv = g[tap] * v[idx] * rsqrtf(sumsq[tap]);
    wf[(((int64_t)gi * Cg + co) * K + tap) * Cg + i] = from_f32<T>(wv);
    wb[(((int64_t)gi * Cg + i) * K + (K - 1 - tap)) * Cg + co] = from_f32<T>(wv);
  }
}

extern "C" int w2v2_weightnorm_pack(const float* g, const float* v, float* sumsq, void* wf, void* wb, int H, int G,
                                    int K, int dtype, void* stream) {
  W2V2_REQUIRE(g && v && sumsq && wf && wb && G > 0 && H % G == 0 && K > 0 && K <= 256, "weightnorm_pack: bad arguments");
  const int Cg = H / G;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((tap_reduce_kernel<false>), dim3(TAP_BLOCKS), dim3(256), 0, st, v, (const float*)nullptr,
                     sumsq + K, H, Cg, K);
  hipLaunchKernelGGL(tap_finalize_kernel, dim3((unsigned)cdiv(K, 128)), dim3(128), 0, st, sumsq + K, sumsq, K);
  const int64_t total = (int64_t)H * Cg * K;
  int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  W2V2_DISPATCH_ACT(dtype, "weightnorm_pack",
    hipLaunchKernelGGL(wn_pack_kernel<AT>, dim3(nb), dim3(256), 0, st, g, v, sumsq, (AT*)wf, (AT*)wb, H, Cg, K););
  W2V2_CHECK_LAUNCH("weightnorm_pack");
  return 0;
}

// w = g_k v / n_k:  dg_k = dot_k / n_k,  dv = g_k/n_k * (dw - v * dot_k / n_k^2),  dot_k = sum dw*v
__global__ void wn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                              const float* __restrict__ sumsq, const float* __restrict__ dwf,
                              const float* __restrict__ dot, float* __restrict__ dg, float* __restrict__ dv, int H,
                              int Cg, int K) {
  const int64_t total = (int64_t)H * Cg * K;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % K);
    const int64_t r = idx / K;
    const int i = (int)(r % Cg), o = (int)(r / Cg);
    const int gi = o / Cg, co = o - gi * Cg;
    const float n2 = sumsq[tap], rn = rsqrtf(n2);
    const float dw = dwf[(((int64_t)gi * K + tap) * Cg + i) * Cg + co];
    dv[idx] = g[tap] * rn * (dw - v[idx] * dot[tap] / n2);
    if (idx < K) dg[idx] = dot[idx] * rsqrtf(sumsq[idx]);
  }
}

extern "C" int w2v2_weightnorm_bwd(const float* g, const float* v, const float* sumsq, const float* dwf, float* dot,
                                   float* dg, float* dv, int H, int G, int K, void* stream) {
  W2V2_REQUIRE(g && v && sumsq && dwf && dot && dg && dv && G > 0 && H % G == 0 && K > 0 && K <= 256,
               "weightnorm_bwd: bad arguments");
  const int Cg = H / G;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((tap_reduce_kernel<true>), dim3(TAP_BLOCKS), dim3(256), 0, st, v, dwf, dot + K, H, Cg, K);
  hipLaunchKernelGGL(tap_finalize_kernel, dim3((unsigned)cdiv(K, 128)), dim3(128), 0, st, dot + K, dot, K);
  const int64_t total = (int64_t)H * Cg * K;
  int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(nb), dim3(256), 0, st, g, v, sumsq, dwf, dot, dg, dv, H, Cg, K);
  W2V2_CHECK_LAUNCH("weightnorm_bwd");
  return 0;
}
