// conv0.hip -- layer 0 of the wav2vec2 feature encoder (HF:302-323, HF:382-419):
// Conv1d(1 -> C, k=10, stride=5, no bias) + GroupNorm(C groups == per-(utterance, channel) statistics
// over time, biased variance) + GELU(erf), fused so the [B, L, C] pre-norm tensor never exists in HBM:
//   pass 1 (stats): conv, per-block partial sum / sum-of-squares per (b, chunk, c), then a tiny
//                   finalize kernel folds the partials in a FIXED order (f64) -> {mean, rstd}:
//                   bitwise deterministic and independent of the batch an utterance sits in.
//                   (16-bit activations, k = 10: no convolution at all -- window moments of the waveform,
//                   see conv0_gram_kernel below.)
//   pass 2 (apply): conv again (10 MAC per output, cheaper than a 9.8 MB/utt round trip),
//                   normalise, GELU, store channels-last in the activation dtype.
// HBM-bound: algorithmic bytes/utt = 2 x 192 KB waveform reads + L*C*sizeof(T) output (9.83 MB bf16).
// The waveform chunk of a block is staged in LDS once and read by broadcast (every thread of the
// block needs the same 10-sample window for a given frame); threads map to adjacent channels so the
// channels-last stores are fully coalesced (512 B per frame per block).
#include "common.h"
#include <stdlib.h>

constexpr int C0_FRAMES = 128;   // frames per block
constexpr int C0_MAXK = 16;

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                    float* __restrict__ partial, const float* __restrict__ mr,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ y, int N,
                                                    int L, int C, int k, int stride, float eps) {
  extern __shared__ float xs[];
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * C0_FRAMES;
  const int nf = min(C0_FRAMES, L - l0);
  const int nsamp = (nf - 1) * stride + k;
  const float* src = wav + (int64_t)b * N + (int64_t)l0 * stride;
  for (int i = threadIdx.x; i < nsamp; i += 256) xs[i] = src[i];
  __syncthreads();

  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[C0_MAXK];
#pragma unroll
    for (int j = 0; j < C0_MAXK; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
    if constexpr (!APPLY) {
      float s1 = 0.f, s2 = 0.f;
      for (int f = 0; f < nf; ++f) {
        const float* xp = xs + f * stride;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < k) acc = fmaf(wr[j], xp[j], acc);
        s1 += acc;
        s2 = fmaf(acc, acc, s2);
      }
      float* pt = partial + (((int64_t)b * gridDim.x + blockIdx.x) * C + c) * 2;
      pt[0] = s1;
      pt[1] = s2;
    } else {
      const float* st = mr + ((int64_t)b * C + c) * 2;
      const float ga = gamma[c] * st[1];
      const float be = beta[c] - st[0] * ga;
      T* dst = y + ((int64_t)b * L + l0) * C + c;
      for (int f = 0; f < nf; ++f) {
        const float* xp = xs + f * stride;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < k) acc = fmaf(wr[j], xp[j], acc);
        dst[(int64_t)f * C] = from_f32<T>(gelu_f(fmaf(acc, ga, be)));
      }
    }
  }
}

// fold the per-chunk partials of one (b, c) in chunk order (f64) -> {mean, rstd}
__global__ void conv0_finalize_kernel(const float* __restrict__ partial, float* __restrict__ mr, int B, int C,
                                      int nchunk, int L, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  double s1 = 0.0, s2 = 0.0;
  for (int j = 0; j < nchunk; ++j) {
    const float* pt = partial + (((int64_t)b * nchunk + j) * C + c) * 2;
    s1 += (double)pt[0];
    s2 += (double)pt[1];
  }
  const double mu = s1 / (double)L;
  const double var = s2 / (double)L - mu * mu;
  mr[(int64_t)i * 2] = (float)mu;
  mr[(int64_t)i * 2 + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
}

// ------------------------------------------------------------------------------ matrix-core variant (bf16 activations)
// The VALU kernel above spends 10 FMA + 10 LDS broadcasts per output on the convolution and stores 2 bytes per
// lane.  Here the convolution of 16 frames x 16 channels is ONE v_mfma_f32_16x16x32_bf16 with f32-class accuracy:
// x = xh + xl and w = wh + wl are split into bf16 pairs and the K = 32 slots carry
//     [ xh(k taps) | xh(k taps) | xl(k taps) | 0 ] . [ wh | wl | wh | 0 ]  =  xh.wh + xh.wl + xl.wh      (3k <= 32)
// (relative error ~2^-16: only the xl.wl term is dropped).  The weight fragments' rows are permuted so that a lane
// ends up with 32 CONSECUTIVE channels of one frame (8 fragments x 4 accumulator registers): the normalise + GELU
// epilogue runs on registers and stores 4 x 16 B per lane, 256 B contiguous per 4 lanes.  What remains is the
// ~25 VALU ops of GELU per output.  The statistics pass is the same MFMA with a sum / sum-of-squares epilogue.
typedef __attribute__((ext_vector_type(8))) __bf16 c0_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c0_f32x4;

__device__ __forceinline__ bf16_t c0_hi(float x) { return f32_to_bf16(x); }
__device__ __forceinline__ bf16_t c0_lo(float x) { return f32_to_bf16(x - bf16_to_f32(f32_to_bf16(x))); }

constexpr int C0_CPB = 5;        // 128-frame chunks per workgroup: amortises the weight / scale fragments

// CW = channels per wave: 128 (four waves per workgroup, 213 VGPRs, two waves per SIMD) or 64 (eight waves, half the
// weight / scale / accumulator registers, four waves per SIMD: the apply pass is ~25 VALU ops per output behind an MFMA and in
// front of a 650 MB store stream, and needs the waves to overlap them)
template <bool APPLY, typename TO, int CW = 128>
__global__ __launch_bounds__(CW == 64 ? 512 : 256, CW == 64 ? 4 : 2) void conv0_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                         float* __restrict__ partial, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, TO* __restrict__ y,
                                                         int N, int L, int C, int k, int stride) {
  // TO = 16-bit output type (bf16 or fp16); the convolution itself is always the split-bf16 product below
  extern __shared__ bf16_t c0_lds[];                 // xh[nmax] | xl[nmax] | one zero slot
  const int nmax = (C0_FRAMES - 1) * stride + k;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;           // operand row / 8-wide k slot; as output: frame row r, channel quad kq
  // LDS offsets of this lane's 8 k slots relative to the first sample of its frame (zero slot for the padding)
  // (packed two to a register, 0xffff = padding: the 64-channel variant runs at the 128-register limit of four waves per SIMD)
  uint32_t xoff2[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ki = kq * 8 + e;
    const int o = ki < k ? ki : ki < 2 * k ? ki - k : ki < 3 * k ? nmax + ki - 2 * k : 0xffff;
    if (e & 1) xoff2[e >> 1] |= (uint32_t)o << 16; else xoff2[e >> 1] = (uint32_t)o;
  }
  const int chunk0 = blockIdx.x * C0_CPB;
  const int nchunk = (L + C0_FRAMES - 1) / C0_FRAMES;
  const int chunk1 = min(chunk0 + C0_CPB, nchunk);

  constexpr int NU = CW / 64, NJ = 4 * NU, NQ = 16 * NU;
  for (int cw0 = 0; cw0 < C; cw0 += 512) {           // uniform trip count: every wave takes part in the staging
    const int cw = cw0 + wave * CW;                  // this wave's CW channels
    const bool active = cw < C;
    // weight fragments: fragment j row rho <-> channel cw + (j >> 2) * 64 + (rho >> 2) * 16 + (j & 3) * 4 + (rho & 3):
    // after the MFMA a lane (frame r, quad kq) holds two runs of 16 consecutive channels, cw + u * 64 + kq * 16 + 0..15
    // (u = 0, 1), and the four kq lanes of a frame cover one whole 128-byte line per run -- see the store below
    c0_bf16x8 wf[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ch = active ? cw + (j >> 2) * 64 + (r >> 2) * 16 + (j & 3) * 4 + (r & 3) : 0;
      const float* wp = w + (int64_t)ch * k;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ki = kq * 8 + e;
        bf16_t v = 0;
        if (ki < k) v = c0_hi(wp[ki]);
        else if (ki < 2 * k) v = c0_lo(wp[ki - k]);
        else if (ki < 3 * k) v = c0_hi(wp[ki - 2 * k]);
        wf[j][e] = __builtin_bit_cast(__bf16, v);
      }
    }
    const int cl = active ? cw + kq * 16 : 0;        // this lane's 32 output channels: cl + (q >> 4) * 64 + (q & 15)
    auto chq = [&](int q) -> int { return cl + (q >> 4) * 64 + (q & 15); };
    float ga[NQ], be[NQ];
    if constexpr (APPLY) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float* st = mr + ((int64_t)b * C + chq(q)) * 2;
        ga[q] = gamma[chq(q)] * st[1];
        be[q] = beta[chq(q)] - st[0] * ga[q];
        // (four at a time: with all the loads of this block in flight at once the 64-channel variant spills two registers,
        //  and a kernel that touches scratch at all pays for its set-up on every dispatch)
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    float s1[NQ], s2[NQ];
    if constexpr (!APPLY) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
    }
#pragma unroll 1
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      const int l0 = chunk * C0_FRAMES;
      const int nf = min(C0_FRAMES, L - l0);
      const int nsamp = (nf - 1) * stride + k;
      const float* src = wav + (int64_t)b * N + (int64_t)l0 * stride;
      __syncthreads();                               // previous chunk fully consumed
      for (int i = threadIdx.x; i <= nmax; i += blockDim.x) {
        const float v = i < nsamp ? src[i] : 0.f;
        if (i < nmax) {
          c0_lds[i] = c0_hi(v);
          c0_lds[nmax + i] = c0_lo(v);
        } else {
          c0_lds[2 * nmax] = 0;                      // the zero slot
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int f0 = 0; f0 < (active ? nf : 0); f0 += 16) {
        // activation fragment: frame f0 + r, k slots kq*8 .. +7; rows past the last frame are exact zeros
        // (frames past the end and padding slots index past the image and are clamped onto the zero slot: one v_min per
        //  element instead of a compare + select with a 64-bit condition each)
        const int sbase = f0 + r < nf ? (f0 + r) * stride : 0x10000;
        c0_bf16x8 xf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int xo = (int)((xoff2[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          const bf16_t v = c0_lds[min(sbase + xo, 2 * nmax)];
          xf[e] = __builtin_bit_cast(__bf16, v);
        }
        c0_f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, c0_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if constexpr (APPLY) {
          // full-line stores (common.h): lanes r and r ^ 8 swap one 16-byte half, then every instruction writes
          // 8 frames x 128 contiguous bytes (the per-lane 64-byte runs of the first version left at 2.2 TB/s)
          const bool lo = r < 8;
          const int fa = f0 + (r & 7);
          TO* dst = y + ((int64_t)b * L + l0 + fa) * C + cl + (lo ? 0 : 8);
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int q = u * 16 + e;
              v[e] = gelu_f(fmaf(acc[q >> 2][q & 3], ga[q], be[q]));
            }
            uint4 da, db;
            halves_to_lines(lo, pack8<TO>(v), pack8<TO>(v + 8), da, db);
            if (fa < nf) store16_wt(dst + u * 64, da);
            if (fa + 8 < nf) store16_wt(dst + (int64_t)8 * C + u * 64, db);
          }
        } else {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const float u = acc[q >> 2][q & 3];
            s1[q] += u;
            s2[q] = fmaf(u, u, s2[q]);
          }
        }
      }
    }
    if constexpr (!APPLY) {
      // fold the 16 frame rows (lanes with equal kq) in a fixed butterfly order, then one lane per quad writes
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1[q] += __shfl_xor(s1[q], o, 64);
          s2[q] += __shfl_xor(s2[q], o, 64);
        }
      }
      if (r == 0 && active) {
        float* pt = partial + ((int64_t)b * gridDim.x + blockIdx.x) * C * 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          pt[2 * chq(q)] = s1[q];
          pt[2 * chq(q) + 1] = s2[q];
        }
      }
    }
  }
}

static const bool g_conv0_mfma = getenv("W2V2_CONV0_VALU") == nullptr;     // A/B switch
static bool conv0_mfma_ok(int C, int k) { return C % 128 == 0 && 3 * k <= 32 && g_conv0_mfma; }

// ------------------------------------------------------------------------------ statistics without the convolution
// The convolution is linear in the waveform, so the per-(utterance, channel) statistics GroupNorm needs are quadratic
// forms of TEN-sample window moments that do not depend on the channel:
//     sum_t y[t,c]   = sum_j w[c,j] S[j],                 S[j]     = sum_t x[t*stride + j]
//     sum_t y[t,c]^2 = sum_j sum_j' w[c,j] w[c,j'] R[j,j'],   R[j,j'] = sum_t x[t*stride + j] x[t*stride + j']
// 65 numbers per utterance (k = 10) instead of a [L, C] convolution (5120 MACs per frame -> 55): the statistics
// pass drops from 100 us to a few.  Products of two f32 are exact in f64 and everything is accumulated in f64 in a fixed
// order (per thread over its frames, the 64 lanes in lane order, blocks in order): deterministic, independent of the
// batch, and more accurate than summing the split-bf16 convolution itself.
template <int K>
__global__ __launch_bounds__(64) void conv0_gram_kernel(const float* __restrict__ wav, double* __restrict__ partial,
                                                        int N, int L, int stride, int fpb) {
  constexpr int NR = K * (K + 1) / 2, NV = K + NR;
  extern __shared__ float xs[];                      // the block's samples: (fpb - 1) * stride + K floats
  __shared__ double red[NV][65];                     // (pitch 65: the fold below reads a row per lane)
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * fpb;                   // frames [f0, f0 + nf) of utterance b; one wave per block
  const int nf = min(fpb, L - f0);
  const float* src = wav + (int64_t)b * N + (int64_t)f0 * stride;
  const int nsamp = nf > 0 ? (nf - 1) * stride + K : 0;
  for (int i = threadIdx.x; i < nsamp; i += 64) xs[i] = src[i];
  __syncthreads();
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  for (int f = threadIdx.x; f < nf; f += 64) {       // lanes `stride` floats apart: conflict-free LDS reads for odd strides
    float x[K];
#pragma unroll
    for (int j = 0; j < K; ++j) x[j] = xs[f * stride + j];
    int idx = K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      acc[j] += (double)x[j];
#pragma unroll
      for (int j2 = j; j2 < K; ++j2) acc[idx++] += (double)x[j] * (double)x[j2];
    }
  }
  // fold over the 64 lanes in lane order through LDS (a shuffle tree costs 2 x 6 ds_bpermute per value: 780 per wave)
#pragma unroll
  for (int i = 0; i < NV; ++i) red[i][threadIdx.x] = acc[i];
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += 64) {
    double v = 0.0;
    for (int j = 0; j < 64; ++j) v += red[i][j];
    partial[((int64_t)b * gridDim.x + blockIdx.x) * NV + i] = v;
  }
}

template <int K>
__global__ __launch_bounds__(256) void conv0_gram_finalize_kernel(const double* __restrict__ partial,
                                                                  const float* __restrict__ w, float* __restrict__ mr,
                                                                  int C, int nblk, int L, float eps) {
  constexpr int NR = K * (K + 1) / 2, NV = K + NR;
  __shared__ double G[NV];
  const int b = blockIdx.x;
  if (threadIdx.x < NV) {
    double v = 0.0;
    for (int j = 0; j < nblk; ++j) v += partial[((int64_t)b * nblk + j) * NV + threadIdx.x];
    G[threadIdx.x] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double wr[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wr[j] = (double)w[c * K + j];
    double s1 = 0.0, s2 = 0.0;
    int idx = K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      s1 += wr[j] * G[j];
#pragma unroll
      for (int j2 = j; j2 < K; ++j2) s2 += (j2 == j ? 1.0 : 2.0) * wr[j] * wr[j2] * G[idx++];
    }
    const double mu = s1 / (double)L;
    const double var = s2 / (double)L - mu * mu;
    mr[((int64_t)b * C + c) * 2] = (float)mu;
    mr[((int64_t)b * C + c) * 2 + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
  }
}
constexpr int C0_GRAM_FRAMES = 640;     // frames per workgroup of the window-moment kernel

extern "C" int w2v2_conv0_workspace_floats(int N, int C, int k, int stride);
static int conv0_check(const char* nm, int B, int N, int C, int k, int stride) {
  W2V2_REQUIRE(B > 0 && C > 0 && k > 0 && k <= C0_MAXK && stride > 0 && N >= k,
               "%s: bad shape B=%d N=%d C=%d k=%d stride=%d", nm, B, N, C, k, stride);
  return 0;
}

extern "C" int w2v2_conv0_stats(const float* wav, const float* w, float* partial, float* mean_rstd, int B, int N,
                                int C, int k, int stride, float eps, void* stream) {
  if (conv0_check("conv0_stats", B, N, C, k, stride)) return -1;
  W2V2_REQUIRE(wav && w && partial && mean_rstd, "conv0_stats: null pointer");
  const int L = (N - k) / stride + 1;
  const int nchunk = (int)cdiv(L, C0_FRAMES);
  dim3 grid((unsigned)nchunk, B);
  const size_t lds = ((size_t)(C0_FRAMES - 1) * stride + k) * sizeof(float);
  hipLaunchKernelGGL((conv0_kernel<float, false>), grid, dim3(256), lds, as_stream(stream), wav, w, partial,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, N, L, C, k,
                     stride, 0.f);
  hipLaunchKernelGGL(conv0_finalize_kernel, dim3((unsigned)cdiv((int64_t)B * C, 256)), dim3(256), 0,
                     as_stream(stream), partial, mean_rstd, B, C, nchunk, L, eps);
  W2V2_CHECK_LAUNCH("conv0_stats");
  return 0;
}

// Same contract as w2v2_conv0_stats, statistics of the split-bf16 matrix-core convolution (what
// w2v2_conv0_apply computes for bf16 outputs); falls back to the exact kernel for shapes the MFMA path does not take.
extern "C" int w2v2_conv0_stats_mfma(const float* wav, const float* w, float* partial, float* mean_rstd, int B, int N,
                                     int C, int k, int stride, float eps, void* stream) {
  if (!conv0_mfma_ok(C, k)) return w2v2_conv0_stats(wav, w, partial, mean_rstd, B, N, C, k, stride, eps, stream);
  if (conv0_check("conv0_stats_mfma", B, N, C, k, stride)) return -1;
  W2V2_REQUIRE(wav && w && partial && mean_rstd, "conv0_stats_mfma: null pointer");
  const int L = (N - k) / stride + 1;
  static const bool no_gram = getenv("W2V2_CONV0_NO_GRAM") != nullptr;        // A/B switch
  if (k == 10 && !no_gram) {
    // statistics from the window moments of the waveform (see conv0_gram_kernel); the workspace (>= 8 floats per frame)
    // holds the f64 partials: 130 floats per block of C0_GRAM_FRAMES frames
    const int nblk = (int)cdiv(L, C0_GRAM_FRAMES);
    double* gp = reinterpret_cast<double*>(partial);
    const size_t lds = ((size_t)(C0_GRAM_FRAMES - 1) * stride + 10) * sizeof(float);
    W2V2_REQUIRE(lds <= 64 * 1024, "conv0_stats_mfma: stride %d too large for the window-moment kernel", stride);
    W2V2_REQUIRE((int64_t)nblk * 130 <= (int64_t)w2v2_conv0_workspace_floats(N, C, k, stride),
                 "conv0_stats_mfma: workspace too small for %d window-moment blocks", nblk);
    hipLaunchKernelGGL((conv0_gram_kernel<10>), dim3((unsigned)nblk, B), dim3(64), lds, as_stream(stream), wav, gp, N, L,
                       stride, C0_GRAM_FRAMES);
    hipLaunchKernelGGL((conv0_gram_finalize_kernel<10>), dim3(B), dim3(256), 0, as_stream(stream), (const double*)gp, w,
                       mean_rstd, C, nblk, L, eps);
    W2V2_CHECK_LAUNCH("conv0_stats_mfma");
    return 0;
  }
  const int nchunk = (int)cdiv(cdiv(L, C0_FRAMES), C0_CPB);      // one partial per workgroup (C0_CPB chunks)
  dim3 grid((unsigned)nchunk, B);
  const size_t lds2 = (2 * ((size_t)(C0_FRAMES - 1) * stride + k) + 8) * sizeof(bf16_t);
  hipLaunchKernelGGL((conv0_mfma_kernel<false, bf16_t>), grid, dim3(256), lds2, as_stream(stream), wav, w, partial,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, N, L, C, k,
                     stride);
  hipLaunchKernelGGL(conv0_finalize_kernel, dim3((unsigned)cdiv((int64_t)B * C, 256)), dim3(256), 0,
                     as_stream(stream), partial, mean_rstd, B, C, nchunk, L, eps);
  W2V2_CHECK_LAUNCH("conv0_stats_mfma");
  return 0;
}

extern "C" int w2v2_conv0_workspace_floats(int N, int C, int k, int stride) {
  const int L = (N - k) / stride + 1;
  return (int)cdiv(L, C0_FRAMES) * C * 2;   // per utterance
}

extern "C" int w2v2_conv0_apply(const float* wav, const float* w, const float* mean_rstd, const float* gamma,
                                const float* beta, void* y, int dtype, int B, int N, int C, int k, int stride,
                                void* stream) {
  if (conv0_check("conv0_apply", B, N, C, k, stride)) return -1;
  W2V2_REQUIRE(wav && w && mean_rstd && gamma && beta && y, "conv0_apply: null pointer");
  const int L = (N - k) / stride + 1;
  dim3 grid((unsigned)cdiv(L, C0_FRAMES), B);
  const size_t lds = ((size_t)(C0_FRAMES - 1) * stride + k) * sizeof(float);
  if ((dtype == W2V2_BF16 || dtype == W2V2_F16) && conv0_mfma_ok(C, k)) {
    const size_t lds2 = (2 * ((size_t)(C0_FRAMES - 1) * stride + k) + 8) * sizeof(bf16_t);
    dim3 grid2((unsigned)cdiv(cdiv(L, C0_FRAMES), C0_CPB), B);
    W2V2_DISPATCH_16(dtype, "conv0_apply",
      if (C % 512 == 0)      // eight waves x 64 channels, four waves per SIMD: 238.6 -> 215.2 us at B = 66 (same box)
        hipLaunchKernelGGL((conv0_mfma_kernel<true, AT, 64>), grid2, dim3(512), lds2, as_stream(stream), wav, w,
                           (float*)nullptr, mean_rstd, gamma, beta, (AT*)y, N, L, C, k, stride);
      else
      hipLaunchKernelGGL((conv0_mfma_kernel<true, AT>), grid2, dim3(256), lds2, as_stream(stream), wav, w,
                         (float*)nullptr, mean_rstd, gamma, beta, (AT*)y, N, L, C, k, stride););
  } else W2V2_DISPATCH_ACT(dtype, "conv0_apply",
    hipLaunchKernelGGL((conv0_kernel<AT, true>), grid, dim3(256), lds, as_stream(stream), wav, w,
                       (float*)nullptr, mean_rstd, gamma, beta, (AT*)y, N, L, C, k, stride, 0.f););
  W2V2_CHECK_LAUNCH("conv0_apply");
  return 0;
}

// HF Conv1d weight [Cout][Cin][k] -> implicit-GEMM B operand [Cout][k][Cin] (K index = tap*Cin + cin,
// matching k consecutive channels-last frames of the input).
template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int k) {
  const int64_t total = (int64_t)Cout * Cin * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int tap = (int)(r % k), co = (int)(r / k);
    out[i] = from_f32<T>(w[((int64_t)co * Cin + ci) * k + tap]);
  }
}

// ------------------------------------------------------------------------------------------ layer-norm family, layer 0
// feat_extract_norm = "layer" (HF:275-299, the "-lv60" / xlsr checkpoints): y[b, l, :] = GELU(LN_c(conv0(x)[b, l, :] + bias)).
// The normalisation runs over the CHANNELS of one frame, so a wave owns whole frames: lane i holds channels 8 i .. 8 i + 7
// (their k taps stay in registers: 8 k floats), the frame's k samples are wave-uniform loads, the two LayerNorm reductions
// are wave reductions, the output is one 16-byte store per lane.  Exact f32 arithmetic up to the output rounding; HBM-bound
// on the [B, L, C] write like the group-norm apply pass.  Forward only (this family's feature extractor runs frozen).
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ out, int B, int N,
                                                            int L, int C, int k, int stride, float eps, int frames_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  const bool on = lane < nch;
  float wr[8][KMAX], bs[8], ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = lane * 8 + e;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) wr[e][j] = (on && j < k) ? w[(int64_t)c * k + j] : 0.f;
    bs[e] = (on && bias != nullptr) ? bias[c] : 0.f;
    ga[e] = on ? gamma[c] : 0.f;
    be[e] = on ? beta[c] : 0.f;
  }
  const int64_t total = (int64_t)B * L;
  const int64_t f0 = ((int64_t)blockIdx.x * 4 + wave) * frames_per_wave;
  for (int64_t f = f0; f < f0 + frames_per_wave && f < total; ++f) {
    const int b = (int)(f / L), l = (int)(f - (int64_t)b * L);
    const float* xp = wav + (int64_t)b * N + (int64_t)l * stride;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = bs[e];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        const float xv = xp[j];                       // wave-uniform address
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaf(wr[e][j], xv, y[e]);
      }
    }
    float sum = 0.f;
    if (on) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += y[e];
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
    if (on) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = y[e] - mean; sq += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    if (on) {
      Vec8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = gelu_f((y[e] - mean) * rstd * ga[e] + be[e]);
      o.store(out + f * C + lane * 8);
    }
  }
}

extern "C" int w2v2_conv0_layernorm_gelu(const float* wav, const float* w, const float* bias, const float* gamma,
                                         const float* beta, void* out, int B, int N, int C, int k, int stride, float eps,
                                         int dtype, void* stream) {
  W2V2_REQUIRE(wav && w && gamma && beta && out && B > 0 && k > 0 && stride > 0 && N >= k, "conv0_layernorm_gelu: bad arguments");
  W2V2_REQUIRE(C % 8 == 0 && C <= 512 && k <= 16, "conv0_layernorm_gelu: C=%d (multiple of 8, <= 512), k=%d (<= 16)", C, k);
  const int L = (N - k) / stride + 1;
  const int64_t total = (int64_t)B * L;
  const int fpw = 8;                                   // frames per wave: amortises the 8 k weight registers' load
  dim3 grid((unsigned)cdiv(total, (int64_t)4 * fpw));
  W2V2_DISPATCH_ACT(dtype, "conv0_layernorm_gelu",
    hipLaunchKernelGGL((conv0_ln_gelu_kernel<AT, 16>), grid, dim3(256), 0, as_stream(stream), wav, w, bias, gamma, beta, (AT*)out,
                       B, N, L, C, k, stride, eps, fpw););
  W2V2_CHECK_LAUNCH("conv0_layernorm_gelu");
  return 0;
}

extern "C" int w2v2_pack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int k, void* stream) {
  W2V2_REQUIRE(w && out && Cout > 0 && Cin > 0 && k > 0, "pack_conv_weight: bad arguments");
  const int64_t total = (int64_t)Cout * Cin * k;
  int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  W2V2_DISPATCH_ACT(dtype, "pack_conv_weight",
    hipLaunchKernelGGL(pack_conv_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), w, (AT*)out, Cout, Cin, k););
  W2V2_CHECK_LAUNCH("pack_conv_weight");
  return 0;
}

// =============================================================================== backward (unfrozen CNN)
// z = gelu(y), y = gamma * yhat + beta, yhat = (u - mu) * rstd, u = conv(x, w)   per (utterance b, channel c)
// Given dz [B,L,C]:  dy = dz * gelu'(y);  dgamma_c = sum dy*yhat;  dbeta_c = sum dy;
//   du = rstd*gamma * (dy - mean_l(dy) - yhat * mean_l(dy*yhat));   dw[c][k] = sum_{b,l} du * x[b, stride*l + k].
// u / yhat / y are RECOMPUTED from the waveform (10 MAC), so nothing of layer 0 is saved by the forward.
//   pass 1: per-(b,c) sums {sum dy, sum dy*yhat}            (f32 atomics into sums[B][C][2], caller zeroes)
//   pass 2: du, accumulate dw (+ dgamma, dbeta once per (b,c)) (f32 atomics, caller zeroes)
template <typename T, int PASS>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                        const float* __restrict__ mr, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const T* __restrict__ dz,
                                                        float* __restrict__ sums, float* __restrict__ dw,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                        int L, int C, int k, int stride) {
  extern __shared__ float xs[];
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * C0_FRAMES;
  const int nf = min(C0_FRAMES, L - l0);
  const int nsamp = (nf - 1) * stride + k;
  const float* src = wav + (int64_t)b * N + (int64_t)l0 * stride;
  for (int i = threadIdx.x; i < nsamp; i += 256) xs[i] = src[i];
  __syncthreads();
  const float invL = 1.0f / (float)L;
  for (int c = threadIdx.x; c < C; c += 256) {
    float wr[C0_MAXK];
#pragma unroll
    for (int j = 0; j < C0_MAXK; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
    const float* st = mr + ((int64_t)b * C + c) * 2;
    const float mu = st[0], rstd = st[1], ga = gamma[c], be = beta[c];
    const T* dzp = dz + ((int64_t)b * L + l0) * C + c;
    if constexpr (PASS == 1) {
      float s1 = 0.f, s2 = 0.f;
      for (int f = 0; f < nf; ++f) {
        const float* xp = xs + f * stride;
        float u = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < k) u = fmaf(wr[j], xp[j], u);
        const float yh = (u - mu) * rstd;
        const float dy = to_f32<T>(dzp[(int64_t)f * C]) * gelu_grad_f(fmaf(yh, ga, be));
        s1 += dy;
        s2 = fmaf(dy, yh, s2);
      }
      float* sp = sums + ((int64_t)b * C + c) * 2;
      unsafeAtomicAdd(sp, s1);
      unsafeAtomicAdd(sp + 1, s2);
    } else {
      const float* sp = sums + ((int64_t)b * C + c) * 2;
      const float m1 = sp[0] * invL, m2 = sp[1] * invL;
      float acc[C0_MAXK];
#pragma unroll
      for (int j = 0; j < C0_MAXK; ++j) acc[j] = 0.f;
      for (int f = 0; f < nf; ++f) {
        const float* xp = xs + f * stride;
        float u = 0.f;
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < k) u = fmaf(wr[j], xp[j], u);
        const float yh = (u - mu) * rstd;
        const float dy = to_f32<T>(dzp[(int64_t)f * C]) * gelu_grad_f(fmaf(yh, ga, be));
        const float du = rstd * ga * (dy - m1 - yh * m2);
#pragma unroll
        for (int j = 0; j < C0_MAXK; ++j)
          if (j < k) acc[j] = fmaf(du, xp[j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < C0_MAXK; ++j)
        if (j < k) unsafeAtomicAdd(dw + c * k + j, acc[j]);
      if (blockIdx.x == 0) {             // once per (b, c)
        unsafeAtomicAdd(dgamma + c, sp[1]);
        unsafeAtomicAdd(dbeta + c, sp[0]);
      }
    }
  }
}

extern "C" int w2v2_conv0_bwd(const float* wav, const float* w, const float* mean_rstd, const float* gamma,
                              const float* beta, const void* dz, float* sums, float* dw, float* dgamma, float* dbeta,
                              int dtype, int B, int N, int C, int k, int stride, void* stream) {
  if (conv0_check("conv0_bwd", B, N, C, k, stride)) return -1;
  W2V2_REQUIRE(wav && w && mean_rstd && gamma && beta && dz && sums && dw && dgamma && dbeta, "conv0_bwd: null pointer");
  const int L = (N - k) / stride + 1;
  dim3 grid((unsigned)cdiv(L, C0_FRAMES), B);
  const size_t lds = ((size_t)(C0_FRAMES - 1) * stride + k) * sizeof(float);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(sums, 0, sizeof(float) * 2 * (size_t)B * C, st) != hipSuccess) W2V2_FAIL("conv0_bwd: memset failed");
#define W2V2_C0B(T_, P_)                                                                                       \
  hipLaunchKernelGGL((conv0_bwd_kernel<T_, P_>), grid, dim3(256), lds, st, wav, w, mean_rstd, gamma, beta,     \
                     (const T_*)dz, sums, dw, dgamma, dbeta, N, L, C, k, stride)
  W2V2_DISPATCH_ACT(dtype, "conv0_bwd", { W2V2_C0B(AT, 1); W2V2_C0B(AT, 2); });
#undef W2V2_C0B
  W2V2_CHECK_LAUNCH("conv0_bwd");
  return 0;
}

// col2im of a strided Conv1d data gradient: col [B*Lout][k*Cin] (rows = output frames, K index = tap*Cin+ci)
// -> dx [B][Lin][Cin], dx[b][r][ci] = sum over (l, tap) with stride*l + tap == r.  Every dx element written.
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ col, T* __restrict__ dx, int B, int Lin, int Lout, int Cin,
                              int k, int stride) {
  const int nch = Cin >> 3;
  const int64_t total = (int64_t)B * Lin * nch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const int64_t row = i / nch;
    const int b = (int)(row / Lin), r = (int)(row - (int64_t)b * Lin);
    float acc[8] = {};
    for (int tap = 0; tap < k; ++tap) {
      const int t = r - tap;
      if (t < 0 || t % stride != 0) continue;
      const int l = t / stride;
      if (l >= Lout) continue;
      Vec8<T> v;
      v.load(col + ((int64_t)b * Lout + l) * ((int64_t)k * Cin) + (int64_t)tap * Cin + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v.v[e];
    }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = acc[e];
    o.store(dx + row * Cin + ch * 8);
  }
}

extern "C" int w2v2_col2im(const void* col, void* dx, int B, int Lin, int Lout, int Cin, int k, int stride, int dtype,
                           void* stream) {
  W2V2_REQUIRE(col && dx && B > 0 && Lin > 0 && Lout > 0 && Cin % 8 == 0 && k > 0 && stride > 0, "col2im: bad arguments");
  const int64_t total = (int64_t)B * Lin * (Cin >> 3);
  int nb = (int)(cdiv(total, 256) > 16384 ? 16384 : cdiv(total, 256));
  W2V2_DISPATCH_ACT(dtype, "col2im",
    hipLaunchKernelGGL(col2im_kernel<AT>, dim3(nb), dim3(256), 0, as_stream(stream), (const AT*)col, (AT*)dx, B, Lin, Lout, Cin, k, stride););
  W2V2_CHECK_LAUNCH("col2im");
  return 0;
}

// gradient of the packed conv weight [Cout][k][Cin] (f32, from the dW GEMM) -> HF layout [Cout][Cin][k], ADDED
__global__ void unpack_conv_grad_kernel(const float* __restrict__ gp, float* __restrict__ g, int Cout, int Cin, int k) {
  const int64_t total = (int64_t)Cout * Cin * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int tap = (int)(r % k), co = (int)(r / k);
    g[((int64_t)co * Cin + ci) * k + tap] += gp[i];
  }
}

extern "C" int w2v2_unpack_conv_grad(const float* gp, float* g, int Cout, int Cin, int k, void* stream) {
  W2V2_REQUIRE(gp && g && Cout > 0 && Cin > 0 && k > 0, "unpack_conv_grad: bad arguments");
  const int64_t total = (int64_t)Cout * Cin * k;
  int nb = (int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
  hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(nb), dim3(256), 0, as_stream(stream), gp, g, Cout, Cin, k);
  W2V2_CHECK_LAUNCH("unpack_conv_grad");
  return 0;
}
