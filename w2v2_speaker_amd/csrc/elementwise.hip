// elementwise.hip -- HBM-bound helpers: dropout, GELU', add, column sums (bias grads), casts,
// SpecAugment mask fill, CLS-token prepend.  16-byte vector accesses, grid-stride loops.
#include "common.h"

static inline int ew_blocks(int64_t nvec) {
  int64_t b = cdiv(nvec, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------- dropout
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p,
                               float inv_keep, uint64_t seed) {
  const int64_t nv = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    Vec8<T> v;
    v.load(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float s0, s1;
      drop_scale2(seed, (uint64_t)(i * 8 + e), p, inv_keep, s0, s1);
      v.v[e] *= s0;
      v.v[e + 1] *= s1;
    }
    v.store(y + i * 8);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (nv << 3) + threadIdx.x;
    y[i] = from_f32<T>(to_f32<T>(x[i]) * drop_scale(seed, (uint64_t)i, p, inv_keep));
  }
}

extern "C" int w2v2_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
  W2V2_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  if (n == 0) return 0;
  const float ik = 1.0f / (1.0f - p);
  W2V2_DISPATCH_ACT(dtype, "dropout",
    hipLaunchKernelGGL(dropout_kernel<AT>, dim3(ew_blocks(n >> 3)), dim3(256), 0, as_stream(stream),
                       (const AT*)x, (AT*)y, n, p, ik, seed););
  W2V2_CHECK_LAUNCH("dropout");
  return 0;
}

// ------------------------------------------------------------------------------------- binary ops
template <typename T, int OP>  // OP 0: y = a * gelu'(b)   OP 1: y = a + b
__global__ void binary_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    Vec8<T> va, vb;
    va.load(a + i * 8);
    vb.load(b + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) va.v[e] = OP == 0 ? va.v[e] * gelu_grad_f(vb.v[e]) : va.v[e] + vb.v[e];
    va.store(y + i * 8);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (nv << 3) + threadIdx.x;
    const float fa = to_f32<T>(a[i]), fb = to_f32<T>(b[i]);
    y[i] = from_f32<T>(OP == 0 ? fa * gelu_grad_f(fb) : fa + fb);
  }
}

template <int OP>
static int launch_binary(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream, const char* nm) {
  W2V2_REQUIRE(a && b && y && n >= 0, "%s: bad arguments", nm);
  if (n == 0) return 0;
  W2V2_DISPATCH_ACT(dtype, nm,
    hipLaunchKernelGGL((binary_kernel<AT, OP>), dim3(ew_blocks(n >> 3)), dim3(256), 0, as_stream(stream),
                       (const AT*)a, (const AT*)b, (AT*)y, n););
  W2V2_CHECK_LAUNCH(nm);
  return 0;
}

extern "C" int w2v2_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream) {
  return launch_binary<0>(dy, pre, dx, n, dtype, stream, "gelu_bwd");
}
// dx = dy * gelu'(pre) over an [M][N] matrix AND out[n] += sum_m dx[m][n] (of the stored, rounded dx) in one pass: the
// positional convolution's bias gradient used to re-read the 15 MB product it had just written (colsum: 16 us).
// 32 column lanes x 8 columns and 8 row lanes per workgroup, rows strided over grid.y, two rows in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const T* __restrict__ dy, const T* __restrict__ pre,
                                                              T* __restrict__ dx, float* __restrict__ out, int M, int N) {
  __shared__ float red[8][32 * 8];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int col = (blockIdx.x * 32 + tx) * 8;
  float acc[8] = {};
  if (col < N) {
    const int step = gridDim.y * 8;
    for (int m0 = blockIdx.y * 8 + ty; m0 < M; m0 += 2 * step) {
      Vec8<T> a[2], b[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u * step;
        if (m < M) { a[u].load(dy + (int64_t)m * N + col); b[u].load(pre + (int64_t)m * N + col); }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = m0 + u * step;
        if (m < M) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = to_f32<T>(from_f32<T>(a[u].v[e] * gelu_grad_f(b[u].v[e])));     // what dx will hold
            a[u].v[e] = v;
            acc[e] += v;
          }
          a[u].store(dx + (int64_t)m * N + col);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
  __syncthreads();
  const int t = ty * 32 + tx;                       // 256 threads fold the 256 columns of the block
  const int c = blockIdx.x * 256 + t;
  if (c < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][t];
    unsafeAtomicAdd(out + c, s);
  }
}

extern "C" int w2v2_gelu_bwd_colsum(const void* dy, const void* pre, void* dx, float* colsum, int M, int N, int dtype,
                                    void* stream) {
  W2V2_REQUIRE(dy && pre && dx && colsum && M >= 0 && N > 0, "gelu_bwd_colsum: bad arguments");
  W2V2_REQUIRE(N % 8 == 0 && (((uintptr_t)dy | (uintptr_t)pre | (uintptr_t)dx) & 15) == 0,
               "gelu_bwd_colsum: N must be a multiple of 8 and the matrices 16-byte aligned");
  if (M == 0) return 0;
  int gy = (int)cdiv(M, 8 * 8);
  if (gy > 160) gy = 160;
  dim3 grid((unsigned)cdiv(N, 32 * 8), gy), block(32, 8);
  W2V2_DISPATCH_ACT(dtype, "gelu_bwd_colsum",
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel<AT>, grid, block, 0, as_stream(stream), (const AT*)dy, (const AT*)pre, (AT*)dx,
                       colsum, M, N););
  W2V2_CHECK_LAUNCH("gelu_bwd_colsum");
  return 0;
}
extern "C" int w2v2_add(const void* x, const void* a, void* y, int64_t n, int dtype, void* stream) {
  return launch_binary<1>(x, a, y, n, dtype, stream, "add");
}

// ------------------------------------------------------------------------------------- colsum
// out[n] += sum_m x[m][n].  blockDim = (64 column lanes, 4 row lanes); each column lane owns VEC
// adjacent columns; rows are strided over grid.y.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                     int M, int N) {
  __shared__ float red[4][64 * VEC];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int col = (blockIdx.x * 64 + tx) * VEC;
  float acc[VEC] = {};
  if (col < N) {
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + ty;
    if constexpr (VEC == 8) {
      // eight rows in flight per thread (the plain loop was a chain of dependent loads: 26 us for the 15 MB pos-conv
      // bias gradient, and more row blocks only add atomics on the same N addresses); same order of additions
      for (; m + 7 * step < M; m += 8 * step) {
        Vec8<T> v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u].load(x + (int64_t)(m + u * step) * ld + col);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[u].v[e];
      }
    }
    if constexpr (VEC == 4) {          // f32: one 16-byte load per thread (8 columns per thread = two half-used requests)
      for (; m + 7 * step < M; m += 8 * step) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(x + (int64_t)(m + u * step) * ld + col);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w; }
      }
    }
    for (; m < M; m += step) {
      if constexpr (VEC == 8) {
        Vec8<T> v;
        v.load(x + (int64_t)m * ld + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v.v[e];
      } else if constexpr (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)m * ld + col);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        acc[0] += to_f32<T>(x[(int64_t)m * ld + col]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[ty][tx * VEC + e] = acc[e];
  __syncthreads();
  if (ty == 0 && col < N) {
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (col + e < N)
        unsafeAtomicAdd(out + col + e, red[0][tx * VEC + e] + red[1][tx * VEC + e] + red[2][tx * VEC + e] +
                                           red[3][tx * VEC + e]);
  }
}

extern "C" int w2v2_colsum(const void* x, int64_t ld, float* out, int M, int N, int dtype, void* stream) {
  W2V2_REQUIRE(x && out && M >= 0 && N > 0 && ld >= N, "colsum: bad arguments");
  if (M == 0) return 0;
  const bool vec = (N % 8 == 0) && (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int VECW = vec ? (dtype == W2V2_F32 ? 4 : 8) : 1;
  int gy = (int)cdiv(M, 4 * 64);
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  dim3 grid((unsigned)cdiv(N, 64 * VECW), gy), block(64, 4);
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "colsum", {
    if (vec) {
      if constexpr (sizeof(AT) == 4) hipLaunchKernelGGL((colsum_kernel<float, 4>), grid, block, 0, st, (const float*)x, ld, out, M, N);
      else hipLaunchKernelGGL((colsum_kernel<AT, 8>), grid, block, 0, st, (const AT*)x, ld, out, M, N);
    }
    else hipLaunchKernelGGL((colsum_kernel<AT, 1>), grid, block, 0, st, (const AT*)x, ld, out, M, N);
  });
  W2V2_CHECK_LAUNCH("colsum");
  return 0;
}

// ------------------------------------------------------------------------------------- zero ranges / mean
// The backward WRITES the large gradients (grouped weight-gradient launch: dW and dbias of every Linear) and ADDS only
// into the small ones (LayerNorm gamma / beta folds, pos-conv bias, masked_spec_embed, ...).  Zeroing the whole 400 MB
// gradient arena every step is therefore mostly wasted HBM writes; one launch over a table of (offset, count) ranges --
// the accumulated tensors plus the slices of the layers LayerDrop skipped -- replaces the memset.
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const int64_t* __restrict__ table) {
  const int64_t off = table[2 * blockIdx.y], n = table[2 * blockIdx.y + 1];
  float* p = base + off;
  // head: up to the first 16-byte boundary; body: float4; tail
  const int64_t head = min(n, (int64_t)((4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3));
  const int64_t nv = (n - head) >> 2;
  float4* pv = reinterpret_cast<float4*>(p + head);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x)
    pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0) {
    if ((int64_t)threadIdx.x < head) p[threadIdx.x] = 0.f;
    const int64_t t0 = head + (nv << 2);
    if (t0 + threadIdx.x < n && threadIdx.x < 4) p[t0 + threadIdx.x] = 0.f;
  }
}

extern "C" int w2v2_zero_ranges(float* base, const int64_t* table, int n_ranges, int blocks_per_range, void* stream) {
  W2V2_REQUIRE(base && table && n_ranges >= 0 && blocks_per_range > 0, "zero_ranges: bad arguments");
  if (n_ranges == 0) return 0;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)blocks_per_range, (unsigned)n_ranges), dim3(256), 0,
                     as_stream(stream), base, table);
  W2V2_CHECK_LAUNCH("zero_ranges");
  return 0;
}

// out[0] = mean(x[0..n)) in a fixed order (one workgroup; n = batch size: the scalar loss of a step)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

extern "C" int w2v2_mean(const float* x, float* out, int n, void* stream) {
  W2V2_REQUIRE(x && out && n > 0, "mean: bad arguments");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, out, n);
  W2V2_CHECK_LAUNCH("mean");
  return 0;
}

// ------------------------------------------------------------------------------------- cast
template <typename T>
__global__ void cast_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t nv = n >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    Vec8<float> v;
    v.load(x + i * 8);
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = v.v[e];
    o.store(y + i * 8);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (nv << 3) + threadIdx.x;
    y[i] = from_f32<T>(x[i]);
  }
}

extern "C" int w2v2_cast(const float* x, void* y, int64_t n, int dtype, void* stream) {
  W2V2_REQUIRE(x && y && n >= 0, "cast: bad arguments");
  if (n == 0) return 0;
  W2V2_DISPATCH_ACT(dtype, "cast",
    hipLaunchKernelGGL(cast_kernel<AT>, dim3(ew_blocks(n >> 3)), dim3(256), 0, as_stream(stream), x, (AT*)y, n););
  W2V2_CHECK_LAUNCH("cast");
  return 0;
}

// ------------------------------------------------------------------------------------- mask fill
template <typename T>
__global__ void mask_fill_kernel(T* __restrict__ h, const uint8_t* __restrict__ mask,
                                 const float* __restrict__ embed, int M, int H) {
  const int nch = H >> 3;
  const int64_t total = (int64_t)M * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / nch), ch = (int)(i - (int64_t)m * nch);
    if (mask[m]) {
      Vec8<float> e;
      e.load(embed + ch * 8);
      Vec8<T> o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.v[k] = e.v[k];
      o.store(h + (int64_t)m * H + ch * 8);
    }
  }
}

// d_embed[c] += sum over masked rows of dh[m][c]; masked rows zeroed.  32 row lanes x 8 column lanes
// (64 columns) per block; per-block LDS reduction, then ONE atomic per column per block (a naive
// per-element atomicAdd serialises ~1300 rows on the same 768 addresses: 0.25 ms).
template <typename T>
__global__ __launch_bounds__(256) void mask_fill_bwd_kernel(T* __restrict__ dh, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ d_embed, int M, int H) {
  __shared__ float red[32][64];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + cl * 8;
  float acc[8] = {};
  if (col < H) {
    for (int m = blockIdx.y * 32 + rl; m < M; m += gridDim.y * 32) {
      if (!mask[m]) continue;
      Vec8<T> v;
      v.load(dh + (int64_t)m * H + col);
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc[k] += v.v[k]; v.v[k] = 0.f; }
      v.store(dh + (int64_t)m * H + col);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][cl * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < H && s != 0.f) unsafeAtomicAdd(d_embed + c, s);
  }
}

// feature-axis SpecAugment (HF:1294-1304): h[b][t][c] = 0 where mask[b][c]; the backward is the same call on dh
template <typename T>
__global__ void mask_feature_kernel(T* __restrict__ h, const uint8_t* __restrict__ mask, int B, int Tn, int H) {
  const int nch = H >> 3;
  const int64_t total = (int64_t)B * Tn * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int b = (int)(row / Tn);
    const uint2 mk = *reinterpret_cast<const uint2*>(mask + (int64_t)b * H + ch * 8);
    if ((mk.x | mk.y) == 0) continue;                 // nothing masked in these 8 channels: row left untouched
    Vec8<T> v;
    v.load(h + row * H + ch * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (((k < 4 ? mk.x >> (8 * k) : mk.y >> (8 * (k - 4))) & 0xff) != 0) v.v[k] = 0.f;
    v.store(h + row * H + ch * 8);
  }
}

extern "C" int w2v2_mask_feature(void* h, const uint8_t* mask, int B, int T, int H, int dtype, void* stream) {
  W2V2_REQUIRE(h && mask && H % 8 == 0 && (reinterpret_cast<uintptr_t>(mask) & 7) == 0, "mask_feature: bad arguments");
  if (B <= 0 || T <= 0) return 0;
  W2V2_DISPATCH_ACT(dtype, "mask_feature",
    hipLaunchKernelGGL(mask_feature_kernel<AT>, dim3(ew_blocks((int64_t)B * T * (H >> 3))), dim3(256), 0,
                       as_stream(stream), (AT*)h, mask, B, T, H););
  W2V2_CHECK_LAUNCH("mask_feature");
  return 0;
}

extern "C" int w2v2_mask_fill(void* h, const uint8_t* mask, const float* embed, int M, int H, int dtype, void* stream) {
  W2V2_REQUIRE(h && mask && embed && H % 8 == 0, "mask_fill: bad arguments");
  if (M <= 0) return 0;
  const int nb = ew_blocks((int64_t)M * (H >> 3));
  W2V2_DISPATCH_ACT(dtype, "mask_fill",
    hipLaunchKernelGGL(mask_fill_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (AT*)h, mask, embed, M, H););
  W2V2_CHECK_LAUNCH("mask_fill");
  return 0;
}

extern "C" int w2v2_mask_fill_bwd(void* dh, const uint8_t* mask, float* d_embed, int M, int H, int dtype, void* stream) {
  W2V2_REQUIRE(dh && mask && d_embed && H % 8 == 0, "mask_fill_bwd: bad arguments");
  if (M <= 0) return 0;
  int gy = (int)cdiv(M, 32 * 2);          // two rows per thread: the mask test + load of a row is a dependent chain
  if (gy < 1) gy = 1;
  dim3 grid((unsigned)cdiv(H, 64), gy);
  W2V2_DISPATCH_ACT(dtype, "mask_fill_bwd",
    hipLaunchKernelGGL(mask_fill_bwd_kernel<AT>, grid, dim3(256), 0, as_stream(stream), (AT*)dh, mask, d_embed, M, H););
  W2V2_CHECK_LAUNCH("mask_fill_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------- CLS token
template <typename T>
__global__ void prepend_kernel(const T* __restrict__ x, T* __restrict__ y, float c, int B, int Tn, int H) {
  const int nch = H >> 3;
  const int64_t total = (int64_t)B * (Tn + 1) * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int b = (int)(row / (Tn + 1)), t = (int)(row - (int64_t)b * (Tn + 1));
    Vec8<T> v;
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] = c;
    } else {
      v.load(x + ((int64_t)b * Tn + (t - 1)) * H + ch * 8);
    }
    v.store(y + row * H + ch * 8);
  }
}

extern "C" int w2v2_prepend_token(const void* x, void* y, float c, int B, int T, int H, int dtype, void* stream) {
  W2V2_REQUIRE(x && y && B > 0 && T > 0 && H % 8 == 0, "prepend_token: bad arguments");
  const int nb = ew_blocks((int64_t)B * (T + 1) * (H >> 3));
  W2V2_DISPATCH_ACT(dtype, "prepend_token",
    hipLaunchKernelGGL(prepend_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)x, (AT*)y, c, B, T, H););
  W2V2_CHECK_LAUNCH("prepend_token");
  return 0;
}

// ------------------------------------------------------------------------------------- batched transpose
// dst_i[C][R] = src_i[R][C]^T for a table of matrices inside one arena (the bf16 weight copies): one launch
// refreshes every pre-transposed weight after the optimiser step, so that the data-gradient GEMMs
// (dX = dY W) run on the same K-contiguous LDS-DMA kernel as the forward products.
template <typename T>
__global__ __launch_bounds__(256) void transpose_many_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                             const int64_t* __restrict__ table) {
  __shared__ T tile[64][66];
  const int64_t* e = table + (int64_t)blockIdx.y * 4;
  const int64_t so = e[0], dof = e[1];
  const int R = (int)e[2], Cc = (int)e[3];
  const int tc = (Cc + 63) / 64, tr = (R + 63) / 64;
  if constexpr (sizeof(T) == 2) {
    // 16-byte path (every encoder weight): 8 elements per load, column gather of eight 2-byte LDS reads, 16-byte
    // store -- the scalar loop below moves 2 bytes per instruction and ran at 1.7 TB/s
    if ((R & 7) == 0 && (Cc & 7) == 0 && (so & 7) == 0 && (dof & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      for (int t = blockIdx.x; t < tc * tr; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
          const int r = i >> 3, cv = (i & 7) * 8;
          if (r0 + r < R && c0 + cv < Cc) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + so + (int64_t)(r0 + r) * Cc + c0 + cv);
            uint32_t* trow = reinterpret_cast<uint32_t*>(&tile[r][cv]);   // 132-byte rows: 4-byte aligned
            trow[0] = v.x; trow[1] = v.y; trow[2] = v.z; trow[3] = v.w;
          }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
          const int c = i >> 3, rv = (i & 7) * 8;
          if (r0 + rv < R && c0 + c < Cc) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              w[e] = (uint32_t)tile[rv + 2 * e][c] | ((uint32_t)tile[rv + 2 * e + 1][c] << 16);
            store16_wt(dst + dof + (int64_t)(c0 + c) * R + r0 + rv, make_uint4(w[0], w[1], w[2], w[3]));
          }
        }
      }
      return;
    }
  }
  for (int t = blockIdx.x; t < tc * tr; t += gridDim.x) {
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      if (r0 + r < R && c0 + c < Cc) tile[r][c] = src[so + (int64_t)(r0 + r) * Cc + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int c = i >> 6, r = i & 63;
      if (r0 + r < R && c0 + c < Cc) dst[dof + (int64_t)(c0 + c) * R + r0 + r] = tile[r][c];
    }
  }
}

extern "C" int w2v2_transpose_many(const void* src, void* dst, const int64_t* table, int n, int blocks_per_matrix,
                                   int dtype, void* stream) {
  W2V2_REQUIRE(src && dst && table && n >= 0 && blocks_per_matrix > 0, "transpose_many: bad arguments");
  if (n == 0) return 0;
  dim3 grid(blocks_per_matrix, n);
  if (dtype == W2V2_BF16 || dtype == W2V2_F16)      // pure data movement: the two 16-bit formats share one instantiation
    hipLaunchKernelGGL(transpose_many_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)src,
                       (bf16_t*)dst, table);
  else if (dtype == W2V2_F32)
    hipLaunchKernelGGL(transpose_many_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)src,
                       (float*)dst, table);
  else
    W2V2_FAIL("transpose_many: bad dtype %d", dtype);
  W2V2_CHECK_LAUNCH("transpose_many");
  return 0;
}
