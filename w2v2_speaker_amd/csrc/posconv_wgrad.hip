// posconv_wgrad.hip -- weight gradient of the grouped positional Conv1d (HF:326-379; K = 128 taps, 16 groups) as a
// correlation on the matrix cores.
//
//   dW[g][j][c][o] = sum_{b,t} dY[b, t, g*Cg + o] * xg[b, g, t + j, c]          (xg = zero-padded group-major input)
//
// As an implicit GEMM this is M = K*Cg = 6144, N = Cg = 48, k = B*T with BOTH operands k-major and an im2col view
// on one side: the generic register-staged transposing kernel ran it at 178 TFLOP/s (0.54 ms, 4 % of the step).
// Here a workgroup owns one group and a block of taps and walks the utterances: dY_b [160 x Cg] and the matching
// xg_b rows [160 + taps x Cg] are brought into LDS once per utterance by LDS-DMA (double-buffered), and EVERY tap of
// the block reads its operand from the same image at a row offset -- `ds_read_b64_tr_b16` transposing reads make a
// tap shift a plain address offset (96-byte rows: any shift is 8-byte aligned).  Accumulators stay in registers
// over the whole batch (one writer per element, fixed order: deterministic), dY fragments are read once per
// utterance and shared by the taps of a wave.
//
// k mapping of a 32-row MFMA step: lane group lg (0..3) takes rows {4 lg .. 4 lg + 3} and {16 + 4 lg ..}; with
// dense rows of 2 Cg bytes the natural {8 lg ..} choice puts lane groups 0 and 1 on the same banks.  The
// contraction index may be permuted freely as long as both operands use the same permutation.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

namespace {

constexpr int PW_TC = 160;                 // time rows per work item (5 MFMA k steps)

union PwFrag {
  struct { short4v a, b; } s;
  frag8_t v;
};

template <int OFF0, int OFF1>
__device__ __forceinline__ void pw_tr_read(PwFrag& f, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.s.a) : "v"(addr), "n"(OFF0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.s.b) : "v"(addr), "n"(OFF1) : "memory");
}
template <int N>
__device__ __forceinline__ void pw_wait(PwFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.s.a), "+v"(f.s.b) : "n"(N));
}

template <int I, int N, typename F>
__device__ __forceinline__ void pw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pw_static_for<I + 1, N>(f);
  }
}

// NF = Cg / 16 fragments per channel dimension, TW = taps per wave (4 waves -> 4 TW taps per workgroup)
template <typename TE, int NF, int TW>
__global__ __launch_bounds__(256) void posconv_wgrad_kernel(const bf16_t* __restrict__ dY,
                                                            const bf16_t* __restrict__ xg, float* __restrict__ dwf,
                                                            int B, int T, int H, int G, int K) {
  constexpr int CG = 16 * NF, CH = 2 * NF, KS = PW_TC / 32;
  constexpr int ROWB = CG * 2;                          // bytes per LDS row
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int tap_blocks = K / (4 * TW);
  const int g = blockIdx.x / tap_blocks, tb = blockIdx.x - g * tap_blocks;
  const int j0 = tb * 4 * TW + wave * TW;               // first tap of this wave
  const int Tp = T + K - 1;
  constexpr int xrows = PW_TC + 4 * TW;                 // xg rows a block of 4 TW taps touches per item
  const int bufbytes = (PW_TC + xrows) * ROWB;
  const int nchunk = (T + PW_TC - 1) / PW_TC, items = B * nchunk;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  // per-lane byte offset inside an image: row 4 lg + li / 4, 8-byte piece li % 4 of the fragment's 32 bytes
  const uint32_t lane_off = (uint32_t)((lg * 4 + (li >> 2)) * ROWB + (li & 3) * 8);

  f32x4 acc[TW][NF][NF];                                // [tap][o fragment][c fragment]
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int c = 0; c < NF; ++c) acc[t][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_item = [&](int it, int s) {
    const int b = it / nchunk, t0 = (it - b * nchunk) * PW_TC;
    char* buf = smem_raw + s * bufbytes;
    // dY rows t0 .. t0 + 159 of utterance b, channels of group g; rows past T are zero
    for (int base = 0; base < PW_TC * CH; base += 256) {
      const int idx = base + tid;
      const int row = idx / CH, ch = idx - row * CH;
      if (idx < PW_TC * CH) {
        if (t0 + row < T) {
          const bf16_t* src = dY + ((int64_t)b * T + t0 + row) * H + g * CG + ch * 8;
          __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(buf + (base + wave * 64) * 16), 16, 0, 0);
        } else {
          *reinterpret_cast<uint4*>(buf + idx * 16) = make_uint4(0, 0, 0, 0);
        }
      }
    }
    // xg rows t0 .. t0 + xrows - 1 (clamped: rows past Tp - 1 only ever meet zero dY rows, but must be finite)
    char* xbuf = buf + PW_TC * ROWB;
    const bf16_t* xb = xg + ((int64_t)b * G + g) * Tp * CG;
    for (int base = 0; base < xrows * CH; base += 256) {
      const int idx = base + tid;
      const int row = idx / CH, ch = idx - row * CH;
      if (idx < xrows * CH) {
        const int tp = min(t0 + tb * 4 * TW + row, Tp - 1);
        __builtin_amdgcn_global_load_lds((gvoid_t*)(xb + (int64_t)tp * CG + ch * 8),
                                         (lvoid_t*)(xbuf + (base + wave * 64) * 16), 16, 0, 0);
      }
    }
  };

  if (items > 0) load_item(0, 0);
#pragma unroll 1
  for (int it = 0; it < items; ++it) {
    const int s = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // item `it` landed everywhere; buffer s^1 is free again
    if (it + 1 < items) load_item(it + 1, s ^ 1);
    const uint32_t dyb = lds0 + (uint32_t)(s * bufbytes) + lane_off;
    const uint32_t xb0 = dyb + (uint32_t)(PW_TC * ROWB) + (uint32_t)(wave * TW * ROWB);

    PwFrag af[KS][NF];                                   // dY fragments of the item: shared by the taps of the wave
    pw_static_for<0, KS>([&](auto ks) {
      pw_static_for<0, NF>([&](auto fo) {
        constexpr int off = decltype(ks)::value * 32 * ROWB + decltype(fo)::value * 32;
        pw_tr_read<off, off + 16 * ROWB>(af[decltype(ks)::value][decltype(fo)::value], dyb);
      });
    });
    // taps x k steps, software-pipelined TWO steps deep (three fragment sets): with one wave per SIMD a transposing
    // read has to cover its ~200-clock latency under the nine MFMAs (144 clocks) of a step, which one step does not
    PwFrag xf[3][NF];
    auto read_x = [&](auto step, PwFrag (&dst)[NF]) {
      constexpr int tw = decltype(step)::value / KS, ks = decltype(step)::value % KS;
      pw_static_for<0, NF>([&](auto fc) {
        constexpr int off = (tw + ks * 32) * ROWB + decltype(fc)::value * 32;
        pw_tr_read<off, off + 16 * ROWB>(dst[decltype(fc)::value], xb0);
      });
    };
    read_x(std::integral_constant<int, 0>{}, xf[0]);
    if constexpr (TW * KS > 1) read_x(std::integral_constant<int, 1>{}, xf[1]);
    pw_static_for<0, KS>([&](auto ks) {
      pw_static_for<0, NF>([&](auto fo) { pw_wait<0>(af[decltype(ks)::value][decltype(fo)::value]); });
    });
    pw_static_for<0, TW * KS>([&](auto step) {
      constexpr int st = decltype(step)::value, tw = st / KS, ks = st % KS, cur = st % 3;
      // LDS returns in order: all but the 2 NF reads of step st+1 (if there is one) must have landed
      pw_static_for<0, NF>([&](auto fc) {
        pw_wait<(st + 1 < TW * KS) ? 2 * NF : 0>(xf[cur][decltype(fc)::value]);
      });
      if constexpr (st + 2 < TW * KS) read_x(std::integral_constant<int, st + 2>{}, xf[(st + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int fo = 0; fo < NF; ++fo)
#pragma unroll
        for (int fc = 0; fc < NF; ++fc)
          acc[tw][fo][fc] = mfma16<TE>(af[ks][fo].v, xf[cur][fc].v, acc[tw][fo][fc]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // D[o][c]: lane li <-> c, registers <-> o = 4 lg + e  ->  dwf[g][(j * Cg + c) * Cg + o], 16 bytes per lane
  float* out = dwf + (int64_t)g * K * CG * CG;
#pragma unroll
  for (int tw = 0; tw < TW; ++tw)
#pragma unroll
    for (int fo = 0; fo < NF; ++fo)
#pragma unroll
      for (int fc = 0; fc < NF; ++fc) {
        const f32x4 v = acc[tw][fo][fc];
        *reinterpret_cast<float4*>(out + ((int64_t)(j0 + tw) * CG + fc * 16 + li) * CG + fo * 16 + lg * 4) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
}

template <typename TE, int NF, int TW>
int launch_posconv_wgrad(const bf16_t* dY, const bf16_t* xg, float* dwf, int B, int T, int H, int G, int K,
                         hipStream_t st) {
  const int CG = 16 * NF;
  const size_t lds = (size_t)2 * (PW_TC + PW_TC + 4 * TW) * CG * 2;
  W2V2_REQUIRE(lds <= 160 * 1024, "posconv_wgrad: K=%d Cg=%d needs %zu bytes of LDS", K, CG, lds);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&posconv_wgrad_kernel<TE, NF, TW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((posconv_wgrad_kernel<TE, NF, TW>), dim3(G * (K / (4 * TW))), dim3(256), lds, st, dY, xg, dwf, B, T,
                     H, G, K);
  return 0;
}

}  // namespace

// dY [B*T, H] bf16 (gradient at the conv output), xg [B, G, T+K-1, Cg] bf16 from w2v2_posconv_regroup(pad_left),
// dwf [G][K*Cg][Cg] f32 (row (j, c), column o) -- the layout w2v2_weightnorm_bwd consumes.  Overwrites dwf.
template <typename TE>
static int posconv_wgrad_geom(const void* dY, const void* xg, float* dwf, int B, int T, int H, int G, int K, hipStream_t st) {
  const int Cg = H / G;
  // Cg = 48: 2 taps per wave = 8 per workgroup -> 256 workgroups at K = 128 (4 taps: 128 workgroups, 175 vs 135 us)
  if (Cg == 48 && K % 8 == 0) return launch_posconv_wgrad<TE, 3, 2>((const bf16_t*)dY, (const bf16_t*)xg, dwf, B, T, H, G, K, st);
  if (Cg == 64 && K % 8 == 0) return launch_posconv_wgrad<TE, 4, 2>((const bf16_t*)dY, (const bf16_t*)xg, dwf, B, T, H, G, K, st);
  if (Cg == 16 && K % 16 == 0) return launch_posconv_wgrad<TE, 1, 4>((const bf16_t*)dY, (const bf16_t*)xg, dwf, B, T, H, G, K, st);
  if (Cg == 32 && K % 16 == 0) return launch_posconv_wgrad<TE, 2, 4>((const bf16_t*)dY, (const bf16_t*)xg, dwf, B, T, H, G, K, st);
  W2V2_FAIL("posconv_wgrad: unsupported geometry Cg=%d K=%d (Cg in {16,32,48,64}, K a multiple of 16)", Cg, K);
}

extern "C" int w2v2_posconv_wgrad(const void* dY, const void* xg, float* dwf, int B, int T, int H, int G, int K,
                                  int dtype, void* stream) {
  W2V2_REQUIRE(dY && xg && dwf && B > 0 && T > 0 && G > 0 && K > 0 && H % G == 0,
               "posconv_wgrad: bad arguments");
  W2V2_REQUIRE((reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(xg) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(dwf) & 15) == 0, "posconv_wgrad: operands must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  int rc = -1;
  W2V2_DISPATCH_16(dtype, "posconv_wgrad", rc = posconv_wgrad_geom<AT>(dY, xg, dwf, B, T, H, G, K, st););
  if (rc) return rc;
  W2V2_CHECK_LAUNCH("w2v2_posconv_wgrad");
  return 0;
}
