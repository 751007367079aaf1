// heads.hip -- speaker classification heads: AAM-softmax (ref: src/optim/loss/aam_softmax.py:50-74)
// and the plain CE head (ref: src/optim/loss/cross_entropy.py:27-31).  The cosine / logit products
// run on the GEMM (EPI_SCALE_RC folds both F.normalize calls into its epilogue); the kernels here do
// the row norms, the margin + scale + softmax + cross-entropy row pass (forward AND the gradient
// wrt the cosines in one sweep over [B, C]) and the F.normalize backward.
#include "common.h"
#include <stdlib.h>

template <typename T>
__global__ __launch_bounds__(256) void row_invnorm_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ inv,
                                                          int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (int64_t)row * ld;
  float s = 0.f;
  if constexpr (sizeof(T) == 4) {
    if ((cols & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      // 16-byte loads, all of a lane's loads independent (the AAM class weights: 37 MB per step)
      const float4* x4 = reinterpret_cast<const float4*>(xr);
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int c = lane; c < (cols >> 2); c += 64) {
        const float4 v = x4[c];
        s = fmaf(v.x, v.x, s); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
      }
      s = (s + s1) + (s2 + s3);
    } else {
      for (int c = lane; c < cols; c += 64) { const float v = to_f32<T>(xr[c]); s = fmaf(v, v, s); }
    }
  } else {
    for (int c = lane; c < cols; c += 64) { const float v = to_f32<T>(xr[c]); s = fmaf(v, v, s); }
  }
  s = wave_sum(s);
  if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);   // F.normalize eps
}

extern "C" int w2v2_row_invnorm(const void* x, int64_t ld, float* inv, int rows, int cols, int dtype, void* stream) {
  W2V2_REQUIRE(x && inv && rows > 0 && cols > 0 && ld >= cols, "row_invnorm: bad arguments");
  dim3 grid((unsigned)cdiv(rows, 4));
  W2V2_DISPATCH_ACT(dtype, "row_invnorm",
    hipLaunchKernelGGL(row_invnorm_kernel<AT>, grid, dim3(256), 0, as_stream(stream), (const AT*)x, ld, inv, rows, cols););
  W2V2_CHECK_LAUNCH("row_invnorm");
  return 0;
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  const float a = sh[0], b = sh[1], c = sh[2], d = sh[3];
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b + c + d);
}

// the same over NW waves, folded in wave order (fixed order: bitwise reproducible)
template <int NW>
__device__ __forceinline__ float block_reduce_n(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// One workgroup per utterance row.  z_c = s*cos_c (c != y), z_y = s*phi(cos_y); loss = lse(z) - z_y.
// dLoss/dcos_c = s * (softmax_c - [c==y]) / B * (c == y ? dphi/dcos : 1).
// (1024 threads: the three passes over a row's ~6000 logits are a handful of dependent load rounds on B = 66 workgroups;
//  with 256 threads they were 24 rounds each and the kernel took 25 us, now 13)
constexpr int AAM_ROW_THREADS = 1024;
template <typename T>
__global__ __launch_bounds__(AAM_ROW_THREADS) void aam_row_kernel(const float* __restrict__ cosv, const int64_t* __restrict__ label,
                                                      float* __restrict__ softmax, float* __restrict__ loss_rows,
                                                      T* __restrict__ dcos_w, T* __restrict__ dcos_x,
                                                      const float* __restrict__ inv_x,
                                                      const float* __restrict__ inv_w, float* __restrict__ rowdot,
                                                      float* __restrict__ colprod, int B, int C, int64_t ldc,
                                                      float margin, float scale, const float* __restrict__ loss_scale,
                                                      float* __restrict__ correct_rows, int easy_margin) {
  constexpr int NT = AAM_ROW_THREADS, NW = NT / 64;
  __shared__ float sh[NW];
  __shared__ int shi[NW];
  const int b = blockIdx.x;
  const int64_t yl = label[b];
  // a label outside [0, C) (data module with more speakers than the head): NaN loss for the row, no gradient, no
  // out-of-range read (torch's CE would raise a device assert)
  const bool bad_label = yl < 0 || yl >= C;
  const int y = bad_label ? 0 : (int)yl;
  const float* cr = cosv + (int64_t)b * ldc;
  const bool plain = margin < 0.f;
  float zy, dphi = 1.0f, sc = plain ? 1.0f : scale;
  {
    const float cy = cr[y];
    if (plain) {
      zy = cy;
    } else {
      const float cos_m = cosf(margin), sin_m = sinf(margin);
      const float th = cosf(3.14159265358979323846f - margin);
      const float mm = sinf(3.14159265358979323846f - margin) * margin;
      const float sine = sqrtf(fminf(fmaxf(1.0f - cy * cy, 0.f), 1.f));
      float phi = cy * cos_m - sine * sin_m;
      // ref: aam_softmax.py:60-63 -- easy_margin: phi where cos > 0, else the cosine itself; otherwise phi where
      // cos > cos(pi - m), else cos - sin(pi - m) * m
      if (easy_margin ? (cy > 0.f) : (cy - th > 0.f)) {
        dphi = cos_m + sin_m * cy / fmaxf(sine, 1e-12f);
      } else {
        phi = easy_margin ? cy : cy - mm;
        dphi = 1.0f;
      }
      zy = phi * scale;
    }
  }
  float mx = -INFINITY;
  int amax = 0;
  for (int c = threadIdx.x; c < C; c += NT) {
    const float z = c == y ? zy : cr[c] * sc;
    if (z > mx) { mx = z; amax = c; }                  // first maximum of this thread's (ascending) columns
  }
  const float mine = mx;
  mx = block_reduce_n<NW>(mx, sh, true);
  if (correct_rows != nullptr) {
    // training accuracy (ref: speaker_recognition_module.py:296-307 torchmetrics.Accuracy on the prediction):
    // arg-max of the softmax = smallest column holding the row maximum
    int cand = (mine == mx) ? amax : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = cand;
    __syncthreads();
    if (threadIdx.x == 0) {
      int best = shi[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) best = min(best, shi[w]);
      correct_rows[b] = (!bad_label && best == y) ? 1.0f : 0.0f;
    }
  }
  float sum = 0.f;
  for (int c = threadIdx.x; c < C; c += NT) sum += __expf((c == y ? zy : cr[c] * sc) - mx);
  sum = block_reduce_n<NW>(sum, sh, false);
  const float inv = 1.0f / sum;
  if (threadIdx.x == 0) loss_rows[b] = bad_label ? __builtin_nanf("") : (mx + __logf(sum)) - zy;
  const float invB = bad_label ? 0.f : (loss_scale ? loss_scale[0] : 1.0f) / (float)B;
  float rd = 0.f;
  for (int c = threadIdx.x; c < C; c += NT) {
    const float cv = cr[c];
    const float p = __expf((c == y ? zy : cv * sc) - mx) * inv;
    softmax[(int64_t)b * ldc + c] = p;
    if (dcos_w != nullptr) {
      float g = (p - (c == y ? 1.0f : 0.0f)) * invB * sc;
      if (c == y) g *= dphi;
      // the two F.normalize scalings are folded into the operands of the two gradient GEMMs
      dcos_w[(int64_t)b * ldc + c] = from_f32<T>(inv_w ? g * inv_w[c] : g);
      if (dcos_x != nullptr) dcos_x[(int64_t)b * ldc + c] = from_f32<T>(inv_x ? g * inv_x[b] : g);
      if (colprod != nullptr) {
        rd = fmaf(g, cv, rd);
        colprod[(int64_t)b * C + c] = g * cv;      // folded over b in a fixed order by the caller: no atomics
      }
    }
  }
  if (rowdot != nullptr) {
    rd = block_reduce_n<NW>(rd, sh, false);
    if (threadIdx.x == 0) rowdot[b] = rd;
  }
}

extern "C" int w2v2_aam_softmax_fwd_bwd(const float* cos, const int64_t* label, float* softmax, float* loss_rows,
                                        void* dcos_w, void* dcos_x, const float* inv_x, const float* inv_w,
                                        float* rowdot, float* colprod, int B, int C, int64_t ldc, float margin,
                                        float scale, const float* loss_scale, float* correct_rows, int easy_margin,
                                        int dtype, void* stream) {
  W2V2_REQUIRE(cos && label && softmax && loss_rows && B > 0 && C > 0 && ldc >= C, "aam_softmax: bad arguments");
  W2V2_DISPATCH_ACT(dtype, "aam_softmax",
    hipLaunchKernelGGL(aam_row_kernel<AT>, dim3(B), dim3(AAM_ROW_THREADS), 0, as_stream(stream), cos, label, softmax,
                       loss_rows, (AT*)dcos_w, (AT*)dcos_x, inv_x, inv_w, rowdot, colprod, B, C, ldc, margin,
                       scale, loss_scale, correct_rows, easy_margin););
  W2V2_CHECK_LAUNCH("aam_softmax");
  return 0;
}

// F.normalize backward: y = x * inv, upstream g = dL/dy:  dx = inv * (g - y * <y, g>) = inv * (g - x * inv * dot)
// where the caller supplies dot[r] = <y_r, g_r> = sum_c dcos[r,c] * cos[r,c] (DESIGN.md "AAM backward").
template <typename TX>
__global__ void normalize_bwd_kernel(const float* __restrict__ g, const TX* __restrict__ x, int64_t ldx,
                                     const float* __restrict__ inv, const float* __restrict__ dot,
                                     float* __restrict__ dx, int rows, int cols, int add) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    const float iv = inv[r];
    const float v = iv * (g[i] - to_f32<TX>(x[(int64_t)r * ldx + c]) * iv * dot[r]);
    dx[i] = add ? dx[i] + v : v;
  }
}

extern "C" int w2v2_normalize_bwd(const float* g, const void* x, int64_t ldx, const float* inv, const float* dot,
                                  float* dx, int rows, int cols, int x_dtype, int add, void* stream) {
  W2V2_REQUIRE(g && x && inv && dot && dx && rows > 0 && cols > 0 && ldx >= cols, "normalize_bwd: bad arguments");
  const int64_t total = (int64_t)rows * cols;
  int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  W2V2_DISPATCH_ACT(x_dtype, "normalize_bwd",
    hipLaunchKernelGGL(normalize_bwd_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), g, (const AT*)x,
                       ldx, inv, dot, dx, rows, cols, add););
  W2V2_CHECK_LAUNCH("normalize_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------ AAM class-weight gradient
// dW of the AAM head in ONE launch (round 5; was: a [C x E x B] product on the generic register-staged GEMM into an f32
// scratch, a column sum, and the normalisation backward = 67 us for a memory-sized job):
//   H1[c][e] = sum_b dcos_x[b][c] * emb[b][e]            (the B = 66 rows are the contraction: an outer-product stream)
//   dot[c]   = sum_b colprod[b][c]                       (fixed order: bitwise reproducible)
//   dW[c][e] = inv_w[c] * (H1[c][e] - W[c][e] * inv_w[c] * dot[c])         (F.normalize backward, see normalize_bwd)
// A workgroup owns CB classes x all E columns (thread = columns tid + 256 j), accumulators in registers, the batch rows
// staged through LDS 64 at a time.  ref: the autograd of src/optim/loss/aam_softmax.py:55 (F.linear of normalised operands).
// Round 5, second form: a workgroup owns 16 classes x 512 columns; the batch rows it contracts over are staged through
// LDS 64 at a time (one burst of 16-byte loads, then no global access in the inner loop), a thread keeps 4 columns x 8
// classes in registers (packed f32 FMAs).  The first form re-read every embedding row from L2 per 6-row chunk and was
// bound by those round trips (65 us).
constexpr int ADW_CB = 16, ADW_EC = 512, ADW_BC = 36;   // 36 staged rows = 39 KiB of LDS: four workgroups per CU, so
// that one's W / dW streaming (74 MB per launch: the HBM floor is 12 us) runs under another's FMAs
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <typename T>
__global__ __launch_bounds__(256) void aam_dw_kernel(const T* __restrict__ dcos_x, int64_t ldc, const T* __restrict__ emb,
                                                     const float* __restrict__ colprod, const float* __restrict__ W,
                                                     const float* __restrict__ inv_w, float* __restrict__ dW, int B,
                                                     int Cn, int E, int chunk) {
  extern __shared__ __attribute__((aligned(16))) char adw_raw[];
  T* es = reinterpret_cast<T*>(adw_raw);                                        // [ADW_BC][ADW_EC]
  float* dc = reinterpret_cast<float*>(adw_raw + (size_t)ADW_BC * ADW_EC * sizeof(T));   // [ADW_BC][ADW_CB]
  float* cdp = dc + ADW_BC * ADW_CB;                                            // [16][ADW_CB]
  float* cd = cdp + 16 * ADW_CB;                                                // [ADW_CB]
  const int c0 = blockIdx.x * ADW_CB, ecol0 = blockIdx.y * ADW_EC, tid = threadIdx.x;
  const int q = tid & 127, h = tid >> 7;            // column quad, class half (wave-uniform)
  const int e0 = ecol0 + 4 * q;
  f32x2_t acc[4][4];                                // [column][class pair of this half]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[j][k] = f32x2_t{0.f, 0.f};
  {     // column dots: thread -> (class k = tid % CB, batch lane tid / CB); lanes folded in a fixed order
    const int k = tid % ADW_CB, bl = tid / ADW_CB;
    float s = 0.f;
    if (c0 + k < Cn)
      for (int b = bl; b < B; b += 256 / ADW_CB) s += colprod[(int64_t)b * Cn + c0 + k];
    cdp[bl * ADW_CB + k] = s;
  }
  constexpr int VPR = ADW_EC * sizeof(T) / 16;      // 16-byte vectors per staged row
  constexpr int EPV = 16 / sizeof(T);               // elements per vector
  for (int b0 = 0; b0 < B; b0 += chunk) {          // chunk <= ADW_BC rows, equal parts of B (host)
    const int nb = min(chunk, B - b0);
    __syncthreads();
    // (all loads of a pass in flight before the first LDS store: written as load-store pairs the fill was a chain of
    // sixteen L2 round trips per thread)
    for (int i0 = tid; i0 < nb * VPR; i0 += 256 * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u, r = i / VPR, vcol = i - r * VPR;
        v[u] = make_uint4(0, 0, 0, 0);
        if (i < nb * VPR && ecol0 + vcol * EPV < E)
          v[u] = *reinterpret_cast<const uint4*>(emb + (int64_t)(b0 + r) * E + ecol0 + vcol * EPV);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u, r = i / VPR, vcol = i - r * VPR;
        if (i < nb * VPR) *reinterpret_cast<uint4*>(es + (int64_t)r * ADW_EC + vcol * EPV) = v[u];
      }
    }
    {
      constexpr int NDV = (ADW_BC * ADW_CB + 255) / 256;
      float dv[NDV];
#pragma unroll
      for (int u = 0; u < NDV; ++u) {
        const int i = tid + 256 * u, r = i / ADW_CB, k = i - r * ADW_CB;
        dv[u] = (r < nb && c0 + k < Cn) ? to_f32<T>(dcos_x[(int64_t)(b0 + r) * ldc + c0 + k]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NDV; ++u)
        if (tid + 256 * u < ADW_BC * ADW_CB) dc[tid + 256 * u] = dv[u];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < nb; ++r) {
      float x[4];
      if constexpr (sizeof(T) == 2) {
        const uint2 w = *reinterpret_cast<const uint2*>(es + r * ADW_EC + 4 * q);
        unpack2<T>(w.x, x[0], x[1]);
        unpack2<T>(w.y, x[2], x[3]);
      } else {
        const float4 w = *reinterpret_cast<const float4*>(es + r * ADW_EC + 4 * q);
        x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
      }
      const float4 d0 = *reinterpret_cast<const float4*>(dc + r * ADW_CB + h * 8);
      const float4 d1 = *reinterpret_cast<const float4*>(dc + r * ADW_CB + h * 8 + 4);
      const f32x2_t d[4] = {{d0.x, d0.y}, {d0.z, d0.w}, {d1.x, d1.y}, {d1.z, d1.w}};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[j][k] = __builtin_elementwise_fma(d[k], f32x2_t{x[j], x[j]}, acc[j][k]);
    }
  }
  __syncthreads();
  if (tid < ADW_CB) {
    float s = 0.f;
    for (int l = 0; l < 256 / ADW_CB; ++l) s += cdp[l * ADW_CB + tid];
    cd[tid] = s;
  }
  __syncthreads();
  if (e0 >= E) return;
  // the eight W rows of this thread in flight together (a per-class load -> use -> store chain was eight HBM round trips)
  float4 wv[8];
  float ivv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = min(c0 + h * 8 + k, Cn - 1);
    wv[k] = *reinterpret_cast<const float4*>(W + (int64_t)c * E + e0);
    ivv[k] = inv_w[c];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + h * 8 + k;
    const float iv = ivv[k], sc = iv * cd[h * 8 + k];
    const float w4[4] = {wv[k].x, wv[k].y, wv[k].z, wv[k].w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = iv * (acc[j][k >> 1][k & 1] - w4[j] * sc);
    if (c < Cn) *reinterpret_cast<float4*>(dW + (int64_t)c * E + e0) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int w2v2_aam_dw(const void* dcos_x, int64_t ldc, const void* emb, const float* colprod, const float* W,
                           const float* inv_w, float* dW, int B, int Cn, int E, int dtype, void* stream) {
  W2V2_REQUIRE(dcos_x && emb && colprod && W && inv_w && dW && B > 0 && Cn > 0 && E > 0 && ldc >= Cn, "aam_dw: bad arguments");
  W2V2_REQUIRE(E % 8 == 0 && ((uintptr_t)emb | (uintptr_t)W | (uintptr_t)dW) % 16 == 0,
               "aam_dw: embedding dim %d must be a multiple of 8 and the operands 16-byte aligned (else: the GEMM path)", E);
  dim3 grid((unsigned)cdiv(Cn, ADW_CB), (unsigned)cdiv(E, ADW_EC));
  hipStream_t st = as_stream(stream);
  W2V2_DISPATCH_ACT(dtype, "aam_dw", {
    const size_t lds = (size_t)ADW_BC * ADW_EC * sizeof(AT) + (size_t)(ADW_BC * ADW_CB + 16 * ADW_CB + ADW_CB) * sizeof(float);
    // the attribute is per DEVICE (a process may drive more than one): set it on every call that needs it, as posconv.hip does
    if (lds > 64 * 1024)
      W2V2_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&aam_dw_kernel<AT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds) == hipSuccess, "aam_dw: cannot raise the dynamic LDS limit to %zu bytes", lds);
    const int nchunk = (int)cdiv(B, ADW_BC), chunk = (int)cdiv(B, nchunk);
    hipLaunchKernelGGL(aam_dw_kernel<AT>, grid, dim3(256), lds, st, (const AT*)dcos_x, ldc, (const AT*)emb, colprod, W, inv_w,
                       dW, B, Cn, E, chunk);
  });
  W2V2_CHECK_LAUNCH("aam_dw");
  return 0;
}

// ------------------------------------------------------------------------------------------ paired-input BCE head
// ref: src/lightning_modules/speaker/wav2vec2_paired_input.py:200-206 (nn.Linear(H, 1) on the CLS token) +
// src/optim/loss/binary_cross_entropy.py:24-40 (binary_cross_entropy_with_logits, mean; prediction = sigmoid).
// Row kernel (one wave per pair): logit, probability, loss row, dlogit = (p - y) / B, demb = dlogit * w.
__global__ __launch_bounds__(64) void bce_row_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                     const float* __restrict__ b, const int64_t* __restrict__ label,
                                                     float* __restrict__ prob, float* __restrict__ loss_rows,
                                                     float* __restrict__ dlogit, float* __restrict__ demb, int B,
                                                     int H, const float* __restrict__ loss_scale) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* e = emb + (int64_t)row * H;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) s = fmaf(e[h], w[h], s);
  s = wave_sum(s) + b[0];
  const int64_t yl = label[row];
  const bool bad_label = yl != 0 && yl != 1;          // NaN loss, zero gradient (see aam_row_kernel)
  const float y = (float)yl;
  const float p = 1.0f / (1.0f + __expf(-s));
  // max(s, 0) - s*y + log(1 + exp(-|s|)): the numerically stable form torch uses
  const float l = fmaxf(s, 0.f) - s * y + log1pf(__expf(-fabsf(s)));
  const float dl = bad_label ? 0.f : (p - y) * (loss_scale ? loss_scale[0] : 1.0f) / (float)B;
  if (lane == 0) {
    prob[row] = p;
    loss_rows[row] = bad_label ? __builtin_nanf("") : l;
    if (dlogit != nullptr) dlogit[row] = dl;
  }
  if (demb != nullptr)
    for (int h = lane; h < H; h += 64) demb[(int64_t)row * H + h] = dl * w[h];
}
// dW[h] = sum_b dlogit[b] emb[b][h] (fixed order), db = sum_b dlogit[b]    (written)
__global__ void bce_wgrad_kernel(const float* __restrict__ emb, const float* __restrict__ dlogit,
                                 float* __restrict__ dw, float* __restrict__ db, int B, int H) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h < H) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) s = fmaf(dlogit[r], emb[(int64_t)r * H + h], s);
    dw[h] = s;
  }
  if (h == 0) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) s += dlogit[r];
    db[0] = s;
  }
}

extern "C" int w2v2_bce_head_fwd_bwd(const float* emb, const float* w, const float* b, const int64_t* label, float* prob,
                                     float* loss_rows, float* dlogit, float* demb, float* dw, float* db, int B, int H,
                                     const float* loss_scale, void* stream) {
  W2V2_REQUIRE(emb && w && b && label && prob && loss_rows && B > 0 && H > 0, "bce_head: bad arguments");
  W2V2_REQUIRE((dlogit == nullptr) == (demb == nullptr) && (demb == nullptr) == (dw == nullptr) &&
                   (dw == nullptr) == (db == nullptr),
               "bce_head: gradient outputs come together");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bce_row_kernel, dim3(B), dim3(64), 0, st, emb, w, b, label, prob, loss_rows, dlogit, demb, B, H,
                     loss_scale);
  if (dw != nullptr)
    hipLaunchKernelGGL(bce_wgrad_kernel, dim3((unsigned)cdiv(H, 256)), dim3(256), 0, st, emb, dlogit, dw, db, B, H);
  W2V2_CHECK_LAUNCH("bce_head");
  return 0;
}
