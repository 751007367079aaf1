// posconv_direct.hip -- the grouped positional convolution (HF:326-379: Conv1d(H, H, K = 128, groups = 16), Cg = 48
// channels per group at w2v2-base) and its data gradient as a DIRECT convolution with the input image resident in LDS.
//
// As an implicit GEMM per group (rounds 1-3: zero-copy overlapping rows through the segmented operand of gemm.hip)
// the product is  [B T] x 48 x 6144  with N = 48: a 128 x 64 tile drags 16 KiB of A per 8 KiB of B through L2 -> LDS for
// every K tile although consecutive A rows overlap in all but 48 of their 6144 elements -- 2.8 GB of L2 -> LDS traffic per
// launch (16 TB/s at 172 us: the launch sat on the load path, 538 TFLOP/s).  Here one workgroup owns one (utterance,
// group, block of 160 frames): the padded image  xg[b, g, t0 .. t0 + 287, 0..47]  (27 KiB, contiguous in HBM) is DMA-ed
// into LDS ONCE, and because the im2col row of frame t is the contiguous run starting at element 48 t of that image,
// an MFMA A fragment (16 frames x 32 k) is a plain ds_read_b128 at byte  96 (t + r) + 2 k  -- conflict-free as it is
// (rows 96 B apart: the 16-byte chunks of the two lane halves of a read interleave).  Only the packed weights
// [48][6144] of the group stream through a three-stage LDS-DMA ring (6 KiB per 64-wide K tile).
//   4 waves; wave w owns frame fragments w, w + 4, w + 8 (3, 3, 2, 2 of the ten) x all three channel fragments;
//   operands swapped (D[n][m]): a lane holds 4 consecutive channels of one frame.
// The k order of every accumulator is that of the implicit GEMM (K tiles ascending, k-steps of 32), so the results are
// BIT-EQUAL to it (tests/test_kernels_gpu.py).  Forward epilogue: bias + GELU (+ the pre-activation for the backward);
// data gradient (the same convolution over the regrouped dY with the flipped weights): + aux.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int PD_MB = 160;                     // frames per workgroup (ten 16-row fragments)
// Two geometries (template parameter CG = channels per group): 48 (w2v2-base, H = 768 / 16 groups) and 64 (wav2vec2-large,
// H = 1024 / 16 groups; round 6).  CG = 64: a frame row of the image is exactly one 128-byte line = one 64-wide K tile (one
// tap), so the A fragments of K tile kt are chunks of row (frame + kt); rows 128 B apart would put the sixteen rows of a
// fragment read on two bank groups, so the 16-byte chunks of a row are XOR-swizzled like a GEMM stage -- applied on the
// SOURCE side of the image DMA and again on the read (the swizzle of row + kt changes with the K tile: two VALU per tile).
template <int CG> struct PdGeom {
  static constexpr int ROWB = CG * 2;                                        // bytes per frame row of the image
  static constexpr int IMG_PIECES = ((PD_MB + 127) * ROWB + 1023) / 1024;    // 1 KiB DMA pieces: 27 / 36
  static constexpr int IMG_BYTES = IMG_PIECES * 1024;
  static constexpr int BSTAGE = CG * 64 * 2;                                 // one K tile of the weights: [CG][64] 16-bit
  static constexpr int LDS = IMG_BYTES + 3 * BSTAGE;
  static constexpr int NJ = CG / 16;                                         // channel fragments
};

__device__ __forceinline__ int pd_swz(int row) { return ((row ^ (row >> 1)) & 3) | (row & 4); }   // gemm_common.h swz()

template <typename TE, int MODE, int CG>   // MODE 0: bias + GELU (aux = pre-activation, may be NULL); 1: + aux
__global__ __launch_bounds__(256) void posconv_direct_kernel(const bf16_t* __restrict__ xg, const bf16_t* __restrict__ w,
                                                             bf16_t* __restrict__ out, bf16_t* __restrict__ aux,
                                                             const float* __restrict__ bias, int B, int Tn, int G,
                                                             int K, int64_t ldc, int mblocks, int64_t xg_elems) {
  using GEO = PdGeom<CG>;
  constexpr int PD_CG = CG, PD_IMG_BYTES = GEO::IMG_BYTES, PD_BSTAGE = GEO::BSTAGE, NJ = GEO::NJ, ROWB = GEO::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = blockIdx.x;
  const int mb = item % mblocks, bg = item / mblocks;
  const int g = bg % G, b = bg / G;
  const int Tp = Tn + K - 1;
  const int t0 = mb * PD_MB;
  const int nk = (K * PD_CG) >> 6;               // K tiles of 64

  // ---- image: 27 KiB starting at frame t0 of xg[b, g] (clamped at the end of the tensor: the frames past T feed only
  // outputs that are never stored)
  {
    const int64_t base = (((int64_t)b * G + g) * Tp + t0) * PD_CG;
    for (int p = wave; p < GEO::IMG_PIECES; p += 4) {
      int64_t off;
      if constexpr (CG == 64) {                   // 8 rows x 8 chunks per piece; the lane lands at chunk (lane & 7) of its row
        const int row = p * 8 + (lane >> 3);
        off = base + (int64_t)row * 64 + (((lane & 7) ^ pd_swz(row)) << 3);
      } else {
        off = base + p * 512 + lane * 8;
      }
      off = off < xg_elems - 8 ? off : xg_elems - 8;
      __builtin_amdgcn_global_load_lds((gvoid_t*)(xg + off), (lvoid_t*)(smem + p * 1024), 16, 0, 0);
    }
  }
  // ---- weights of group g: [48][K * 48], K-contiguous; stage = rows x 64 k, 128-byte rows, chunk-swizzled
  const bf16_t* wg = w + (int64_t)g * PD_CG * K * PD_CG;
  const int c8 = lane & 7, r8 = lane >> 3;
  // waves 0..2 issue two 8-row pieces each per stage, wave 3 none
  int boff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + r8;
    boff[j] = (row < PD_CG ? row : PD_CG - 1) * (K * PD_CG) + ((c8 ^ pd_swz(row)) << 3);
  }
  const bool loader = wave * 16 < PD_CG;         // waves 0..2 (48 channels) / 0..3 (64) issue two 8-row pieces per stage
  auto issue = [&](int kt) {
    if (loader) {
      char* dst = smem + PD_IMG_BYTES + (kt % 3) * PD_BSTAGE + (wave * 2) * 1024;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(wg + boff[j] + kt * 64), (lvoid_t*)(dst + j * 1024), 16, 0, 0);
    }
  };
  issue(0);
  if (nk > 1) issue(1);

  const int fr = lane & 15, kg = lane >> 4;
  const int nmf = (wave < 2) ? 3 : 2;            // frame fragments of this wave: w, w + 4, (w + 8)
  // A: byte offset of (frame 16 i + fr, k chunk kg) in the image; B: fragment j, row 16 j + fr
  const int aoff = fr * ROWB + kg * 16;
  const int sw = pd_swz(fr);
  f32x4 acc[3][NJ];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt (and, the first time, the image) landed: at most the pieces of stage kt + 1 stay in flight
    if (loader) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(kt + 2);              // into the buffer every wave finished reading before this barrier
    const char* bs = smem + PD_IMG_BYTES + (kt % 3) * PD_BSTAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t bf[NJ], af[3];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        bf[j] = *reinterpret_cast<const frag8_t*>(bs + (j * 16 + fr) * 128 + (((kk * 4 + kg) ^ sw) << 4));
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nmf) {
          if constexpr (CG == 64)       // row (frame + kt) of the image, logical chunk kk * 4 + kg, swizzled by that row
            af[i] = *reinterpret_cast<const frag8_t*>(smem + ((wave + 4 * i) * 16 + fr + kt) * 128 +
                                                      (((kk * 4 + kg) ^ pd_swz(fr + kt)) << 4));
          else
            af[i] = *reinterpret_cast<const frag8_t*>(smem + (wave + 4 * i) * (16 * 96) + aoff + kt * 128 + kk * 64);
        }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < nmf) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<TE>(bf[j], af[i], acc[i][j]);
        }
    }
  }

  // ---- epilogue: lane (fr, kg) holds channels 16 j + 4 kg .. + 3 of frame 16 (w + 4 i) + fr.  Lanes kg and kg ^ 1
  // exchange one 8-byte piece per block pair (v_permlane16_swap, see attention.hip store_row4x4), so every store is a
  // 16-byte write-through one: blocks 0 / 1 go to the even / odd lane, block 2 (its partner piece is a dummy) to the even one
  const bool odd = (kg & 1) != 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= nmf) continue;                          // (wave-uniform)
    const int t = t0 + (wave + 4 * i) * 16 + fr;
    const bool valid = t < Tn;                       // (the same for the two lanes of a pair: it depends on fr only)
    const int64_t rowoff = ((int64_t)b * Tn + (valid ? t : 0)) * ldc + g * PD_CG;
    uint2 ow[NJ], pw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = j * 16 + kg * 4;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (MODE == 0) {
        float pre[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pre[e] = v[e] + bias[g * PD_CG + n + e];
          v[e] = gelu_f(pre[e]);
        }
        pw[j].x = pack2<TE>(pre[0], pre[1]);
        pw[j].y = pack2<TE>(pre[2], pre[3]);
      } else {
        const uint2 a2 = *reinterpret_cast<const uint2*>(aux + rowoff + n);
        float a0, a1, a2f, a3;
        unpack2<TE>(a2.x, a0, a1);
        unpack2<TE>(a2.y, a2f, a3);
        v[0] += a0; v[1] += a1; v[2] += a2f; v[3] += a3;
      }
      ow[j].x = pack2<TE>(v[0], v[1]);
      ow[j].y = pack2<TE>(v[2], v[3]);
    }
    auto store48 = [&](bf16_t* dst, const uint2 (&w)[NJ]) {
      const auto sx = __builtin_amdgcn_permlane16_swap(w[0].x, w[1].x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(w[0].y, w[1].y, false, false);
      if constexpr (NJ == 4) {          // 64 channels: two block pairs, every lane stores one 16-byte piece of each
        const auto tx = __builtin_amdgcn_permlane16_swap(w[2].x, w[3].x, false, false);
        const auto ty = __builtin_amdgcn_permlane16_swap(w[2].y, w[3].y, false, false);
        if (valid) {
          store16_wt(dst + rowoff + (odd ? 16 : 0) + (kg & 2) * 4, make_uint4(sx[0], sy[0], sx[1], sy[1]));
          store16_wt(dst + rowoff + 32 + (odd ? 16 : 0) + (kg & 2) * 4, make_uint4(tx[0], ty[0], tx[1], ty[1]));
        }
      } else {
        const auto tx = __builtin_amdgcn_permlane16_swap(w[2].x, 0u, false, false);
        const auto ty = __builtin_amdgcn_permlane16_swap(w[2].y, 0u, false, false);
        if (valid) {
          store16_wt(dst + rowoff + (odd ? 16 : 0) + (kg & 2) * 4, make_uint4(sx[0], sy[0], sx[1], sy[1]));
          if (!odd) store16_wt(dst + rowoff + 32 + (kg & 2) * 4, make_uint4(tx[0], ty[0], tx[1], ty[1]));
        }
      }
    };
    if constexpr (MODE == 0) {
      if (aux != nullptr) store48(aux, pw);
    }
    store48(out, ow);
  }
}

extern "C" int w2v2_posconv_direct(const void* xg, const void* w, void* out, void* aux, const float* bias, int B, int T,
                                   int G, int Cg, int K, int64_t ldc, int mode, int dtype, void* stream) {
  W2V2_REQUIRE(xg && w && out && B > 0 && T > 0 && G > 0 && K > 0, "posconv_direct: bad arguments");
  W2V2_REQUIRE((Cg == 48 || Cg == 64) && K == 128, "posconv_direct: built for 48 (w2v2-base) or 64 (wav2vec2-large) channels "
               "per group and 128 taps; got Cg=%d K=%d -- use the implicit GEMM", Cg, K);
  W2V2_REQUIRE(mode == 0 ? bias != nullptr : (mode == 1 && aux != nullptr), "posconv_direct: mode 0 needs bias, mode 1 aux");
  W2V2_REQUIRE(ldc % 8 == 0 && ldc >= G * Cg, "posconv_direct: ldc must be a multiple of 8 and >= G * Cg");
  W2V2_REQUIRE(((uintptr_t)xg | (uintptr_t)w | (uintptr_t)out | (uintptr_t)aux) % 16 == 0,
               "posconv_direct: xg, w, out and aux must be 16-byte aligned (16-byte epilogue stores, LDS-DMA source rows)");
  W2V2_REQUIRE(dtype == W2V2_BF16 || dtype == W2V2_F16, "posconv_direct: needs a 16-bit activation dtype (got %d)", dtype);
  const int mblocks = (int)cdiv(T, PD_MB);
  const int64_t xg_elems = (int64_t)B * G * (T + K - 1) * Cg;
  W2V2_REQUIRE(xg_elems >= 8, "posconv_direct: input too small");
  dim3 grid((unsigned)(B * G * mblocks));
  hipStream_t st = as_stream(stream);
#define PD_LAUNCH(TE_, MODE_, CG_)                                                                                  \
  do {                                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&posconv_direct_kernel<TE_, MODE_, CG_>),                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, PdGeom<CG_>::LDS);                          \
    hipLaunchKernelGGL((posconv_direct_kernel<TE_, MODE_, CG_>), grid, dim3(256), PdGeom<CG_>::LDS, st,               \
                       (const bf16_t*)xg, (const bf16_t*)w, (bf16_t*)out, (bf16_t*)aux, bias, B, T, G, K, ldc, mblocks, \
                       xg_elems);                                                                                    \
  } while (0)
#define PD_LAUNCH_T(TE_)                                                                                             \
  do {                                                                                                               \
    if (Cg == 48) { if (mode == 0) PD_LAUNCH(TE_, 0, 48); else PD_LAUNCH(TE_, 1, 48); }                               \
    else { if (mode == 0) PD_LAUNCH(TE_, 0, 64); else PD_LAUNCH(TE_, 1, 64); }                                       \
  } while (0)
  if (dtype == W2V2_BF16) PD_LAUNCH_T(bf16_t); else PD_LAUNCH_T(f16_t);
#undef PD_LAUNCH_T
#undef PD_LAUNCH
  W2V2_CHECK_LAUNCH("posconv_direct");
  return 0;
}
