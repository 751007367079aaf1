// skinny.hip -- exact-f32 linear layers over a HANDFUL of rows (one row per utterance): the squeeze-excitation
// bottleneck of ECAPA (ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85 -> speechbrain SEBlock: mean_t ->
// Conv1d(C, S, 1) -> ReLU -> Conv1d(S, C, 1) -> sigmoid), the ECAPA embedding layer fc (lin_neurons) and the
// hidden_fc_layers_out stack of the wav2vec2 head (ref: src/lightning_modules/speaker/wav2vec2_fc.py:108-141).
//
// With B ~ 66 rows these products have 2..40 output tiles of a 64x64 GEMM and a long K: a tiled GEMM runs them on
// a few CUs with one staged K step per iteration (latency bound, ~25 us each, and the step has ~20 of them).  Here
// the WEIGHT matrix is the streamed operand: every wave owns one weight row (fwd / dW) or one 64-wide column strip
// (dx), the batch rows ride along in registers, and the activation (or its derivative) is applied on the way in or
// out, so one SE block is 2 + 4 launches instead of 4 + 9.  All sums are f32 in a fixed order (deterministic).
//
//   act: 0 none, 1 relu, 2 sigmoid.  Backward kernels take the forward OUTPUT y and form dy' = dy * act'(y).
#include "common.h"

constexpr int SK_BT = 16;   // batch rows per workgroup of the forward kernel
constexpr int SK_NT = 4;    // weight rows per workgroup of the forward kernel
constexpr int SK_BX = 8;    // batch rows per workgroup of the dx kernel
constexpr int SK_NC = 1024; // dy' columns staged per LDS chunk of the dx kernel
constexpr int SK_XW = 8;    // waves (n slices) of the dx kernel

__device__ __forceinline__ float sk_act(float v, int act) {
  return act == 1 ? fmaxf(v, 0.f) : act == 2 ? 1.0f / (1.0f + __expf(-v)) : v;
}
__device__ __forceinline__ float sk_dact(float dy, float y, int act) {
  return act == 1 ? (y > 0.f ? dy : 0.f) : act == 2 ? dy * y * (1.0f - y) : dy;
}

// Sum 64 per-lane accumulators over the 64 lanes of a wave with 63 exchanges instead of 64 x 6: at every step a lane
// keeps one half of its remaining accumulators and hands the other half to its partner (lane ^ M), so the register
// count halves while the partner distance halves.  Lane L ends up with the wave total of accumulator L.
template <int M, int NN> struct sk_fold {
  static __device__ __forceinline__ void run(float* acc, int lane) {
    const bool hi = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < NN / 2; ++i) {
      const float send = hi ? acc[i] : acc[i + NN / 2];
      const float keep = hi ? acc[i + NN / 2] : acc[i];
      acc[i] = keep + __shfl_xor(send, M, 64);
    }
    sk_fold<M / 2, NN / 2>::run(acc, lane);
  }
};
template <int NN> struct sk_fold<0, NN> {
  static __device__ __forceinline__ void run(float*, int) {}
};

// y[b][n] = act(sum_k x[b][k] W[n][k] + bias[n]);   grid (ceil(N/4), ceil(B/16)): one workgroup = 4 weight rows x 16
// batch rows, its 256 threads span 1024 consecutive k per iteration (the waves split K: ceil(K/1024) dependent
// rounds of 20 independent 16-byte loads), then the wave fold above and a 4-way LDS fold in wave order
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int N, int K, int act) {
  static_assert(SK_NT * SK_BT == 64, "one accumulator per lane after the fold");
  __shared__ float red[4][SK_NT * SK_BT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * SK_NT, b0 = blockIdx.y * SK_BT;
  float acc[SK_NT * SK_BT] = {};
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    float4 wv[SK_NT];
#pragma unroll
    for (int j = 0; j < SK_NT; ++j)
      wv[j] = *reinterpret_cast<const float4*>(W + (int64_t)min(n0 + j, N - 1) * K + k);   // clamped rows: dropped
#pragma unroll
    for (int i = 0; i < SK_BT; ++i) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)min(b0 + i, B - 1) * K + k);
#pragma unroll
      for (int j = 0; j < SK_NT; ++j)
        acc[j * SK_BT + i] =
            fmaf(wv[j].x, xv.x, fmaf(wv[j].y, xv.y, fmaf(wv[j].z, xv.z, fmaf(wv[j].w, xv.w, acc[j * SK_BT + i]))));
    }
  }
  sk_fold<32, 64>::run(acc, lane);
  red[w][lane] = acc[0];
  __syncthreads();
  if (threadIdx.x < SK_NT * SK_BT) {
    const int j = threadIdx.x / SK_BT, i = threadIdx.x % SK_BT;
    const int n = n0 + j, b = b0 + i;
    if (n < N && b < B) {
      const float s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
      y[(int64_t)b * N + n] = sk_act(s + (bias ? bias[n] : 0.f), act);
    }
  }
}

// dx[b][k] = sum_n dy'[b][n] W[n][k];   grid (ceil(K/64), ceil(B/8)), 8 waves: lane = column k, wave w takes the
// rows n = w mod 8 (8 coalesced 256-byte loads in flight), the 8 wave results are folded through LDS in wave order
__global__ __launch_bounds__(512) void skinny_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ W, float* __restrict__ dx, int B,
                                                           int N, int K, int act) {
  __shared__ float dyl[SK_BX][SK_NC];
  __shared__ float red[SK_XW][SK_BX][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * SK_BX;
  const bool ok = k < K;
  float acc[SK_BX] = {};
  for (int n0 = 0; n0 < N; n0 += SK_NC) {
    const int nc = min(SK_NC, N - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < SK_BX * nc; i += 512) {
      const int r = i / nc, c = i - r * nc;
      float v = 0.f;
      if (b0 + r < B) {
        const int64_t o = (int64_t)(b0 + r) * N + n0 + c;
        v = sk_dact(dy[o], act ? y[o] : 0.f, act);
      }
      dyl[r][c] = v;
    }
    __syncthreads();
    if (ok) {
      const float* wp = W + (int64_t)n0 * K + k;
      int n = w;
      for (; n + 7 * SK_XW < nc; n += 8 * SK_XW) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = wp[(int64_t)(n + u * SK_XW) * K];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < SK_BX; ++i) acc[i] = fmaf(dyl[i][n + u * SK_XW], wv[u], acc[i]);
      }
      for (; n < nc; n += SK_XW) {
        const float wv = wp[(int64_t)n * K];
#pragma unroll
        for (int i = 0; i < SK_BX; ++i) acc[i] = fmaf(dyl[i][n], wv, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < SK_BX; ++i) red[w][i][lane] = acc[i];
  __syncthreads();
  if (w < SK_BX && ok && b0 + w < B) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < SK_XW; ++g) s += red[g][w][lane];
    dx[(int64_t)(b0 + w) * K + k] = s;
  }
}

// dW[n][k] (+)= sum_b dy'[b][n] x[b][k],  dbias[n] (+)= sum_b dy'[b][n];   grid (ceil(K/256), ceil(N/4)): a wave owns
// weight row n and 256 columns; dy'[.][n] is fetched 64 rows at a time (one row per lane) and broadcast from
// registers, so the b loop holds only independent x loads (8 in flight)
__global__ __launch_bounds__(256) void skinny_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, float* __restrict__ dW,
                                                           float* __restrict__ dbias, int B, int N, int K, int act,
                                                           int accumulate) {
  const int lane = threadIdx.x & 63;
  const int n = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
  if (n >= N) return;
  const int k = min(blockIdx.x * 256 + lane * 4, K - 4);     // lanes past K recompute the last columns, no store
  const bool ok = blockIdx.x * 256 + lane * 4 < K;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int b0 = 0; b0 < B; b0 += 64) {
    float dl = 0.f;
    if (b0 + lane < B) {
      const int64_t o = (int64_t)(b0 + lane) * N + n;
      dl = sk_dact(dy[o], act ? y[o] : 0.f, act);
    }
    sb += wave_sum(dl);
    const int nb = min(64, B - b0);
    const float* xp = x + (int64_t)b0 * K + k;
    int i = 0;
    for (; i + 8 <= nb; i += 8) {
      float4 xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4*>(xp + (int64_t)(i + u) * K);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), i + u));
        acc.x = fmaf(d, xv[u].x, acc.x);
        acc.y = fmaf(d, xv[u].y, acc.y);
        acc.z = fmaf(d, xv[u].z, acc.z);
        acc.w = fmaf(d, xv[u].w, acc.w);
      }
    }
    for (; i < nb; ++i) {
      const float4 xv = *reinterpret_cast<const float4*>(xp + (int64_t)i * K);
      const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl), i));
      acc.x = fmaf(d, xv.x, acc.x);
      acc.y = fmaf(d, xv.y, acc.y);
      acc.z = fmaf(d, xv.z, acc.z);
      acc.w = fmaf(d, xv.w, acc.w);
    }
  }
  if (ok) {
    float4* dst = reinterpret_cast<float4*>(dW + (int64_t)n * K + k);
    if (accumulate) {
      const float4 o = *dst;
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *dst = acc;
  }
  if (dbias && blockIdx.x == 0 && lane == 0) dbias[n] = accumulate ? dbias[n] + sb : sb;
}

// ------------------------------------------------------------------------------------------ C ABI
static bool sk_ok(const void* p, int K) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && K % 4 == 0; }

extern "C" int w2v2_skinny_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int N, int K,
                                      int act, void* stream) {
  W2V2_REQUIRE(x && W && y && B > 0 && N > 0 && K > 0 && act >= 0 && act <= 2, "skinny_linear_fwd: bad arguments");
  W2V2_REQUIRE(sk_ok(x, K) && sk_ok(W, K), "skinny_linear_fwd: K %% 4 == 0 and 16-byte aligned operands");
  dim3 grid((unsigned)cdiv(N, SK_NT), (unsigned)cdiv(B, SK_BT));
  hipLaunchKernelGGL(skinny_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, W, bias, y, B, N, K, act);
  W2V2_CHECK_LAUNCH("skinny_linear_fwd");
  return 0;
}

extern "C" int w2v2_skinny_linear_bwd_x(const float* dy, const float* y, const float* W, float* dx, int B, int N, int K,
                                        int act, void* stream) {
  W2V2_REQUIRE(dy && W && dx && B > 0 && N > 0 && K > 0 && act >= 0 && act <= 2 && (act == 0 || y),
               "skinny_linear_bwd_x: bad arguments");
  dim3 grid((unsigned)cdiv(K, 64), (unsigned)cdiv(B, SK_BX));
  hipLaunchKernelGGL(skinny_bwd_x_kernel, grid, dim3(512), 0, as_stream(stream), dy, y, W, dx, B, N, K, act);
  W2V2_CHECK_LAUNCH("skinny_linear_bwd_x");
  return 0;
}

extern "C" int w2v2_skinny_linear_bwd_w(const float* dy, const float* y, const float* x, float* dW, float* dbias, int B,
                                        int N, int K, int act, int accumulate, void* stream) {
  W2V2_REQUIRE(dy && x && dW && B > 0 && N > 0 && K > 0 && act >= 0 && act <= 2 && (act == 0 || y),
               "skinny_linear_bwd_w: bad arguments");
  W2V2_REQUIRE(sk_ok(x, K) && sk_ok(dW, K), "skinny_linear_bwd_w: K %% 4 == 0 and 16-byte aligned operands");
  dim3 grid((unsigned)cdiv(K, 256), (unsigned)cdiv(N, 4));
  hipLaunchKernelGGL(skinny_bwd_w_kernel, grid, dim3(256), 0, as_stream(stream), dy, y, x, dW, dbias, B, N, K, act,
                     accumulate);
  W2V2_CHECK_LAUNCH("skinny_linear_bwd_w");
  return 0;
}
