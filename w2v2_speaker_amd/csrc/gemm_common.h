// gemm_common.h -- what the GEMM translation units share: the kernel argument block, the LDS chunk swizzles, the
// epilogues (per-lane, LDS-staged and full-line register forms) and the tile bookkeeping.  The kernels live in
//   gemm.hip        host dispatch (w2v2_gemm), the register-staged generic kernel, the 128x128 LDS-DMA kernel, exact f32
//   gemm_ring.hip   256x128x64 three-stage LDS-DMA ring (N <= 2304 products, two-term weights, deferred stores)
//   gemm_phased.hip 256x256x64 phased kernel (anti-phase wave groups: FFN1, dH, conv stack)
//   gemm_f32.hip    exact-f32 products on v_mfma_f32_32x32x2_f32 (parity mode, ECAPA-TDNN at `precision: 32`)
// (one file per kernel family keeps a rebuild after an edit to ~1 minute instead of four)
#pragma once
#include "common.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct OpDev {
  const void* ptr;
  int64_t ld, seg_len, seg_stride;
  int trans;
  int vec_ok;  // 16-byte vector loads legal (alignment of ptr/ld/strides)
};

struct GemmArgs {
  int M, N, K;
  int epilogue, split_k, atomic;
  int tiles_m, tiles_n;
  int k_per_split;
  OpDev A, B;
  int64_t a_s0, a_s1, b_s0, b_s1;
  int batch_inner;
  void* C;
  int64_t ldc, c_s0, c_s1;
  void* aux;
  int64_t ldaux, aux_s0, aux_s1;
  const float* bias;
  int64_t bias_s1;
  const float* row_scale;
  const float* col_scale;
  float alpha;
  int c_vec_ok;    // 8-element vector stores to C legal
  int aux_vec_ok;  // 8-element vector access to aux legal
  int defer_ok;    // 256x128 ring kernel: stores of a tile may be issued under the next tile's main loop
  int wt_stores;   // full-line epilogue: 1 = write-through (sc1) stores, 0 = plain write-back stores (see w2v2_gemm)
  int late_dma;    // phased kernel: DMA pieces of a phase issued between its MFMAs (host-side choice, see w2v2_gemm)
  // two-term weights (w2v2_hip.h): tiles with n0 >= n_ext_from run k_ext more K steps against B + b_lo_off
  int k_ext, n_ext_from;
  int64_t b_lo_off;
  int xcd_tiles;   // gemm_f32_dma: 1 = every XCD takes a contiguous run of tiles (set by w2v2_launch_gemm_f32)
};

__device__ __forceinline__ int64_t outer_off(const OpDev& o, int64_t idx) {
  if (o.seg_len > 0) {
    const int64_t q = idx / o.seg_len;
    return q * o.seg_stride + (idx - q * o.seg_len) * o.ld;
  }
  return idx * o.ld;
}

// depends on (row & 15) only: every 16-row MFMA fragment of a tile shares one per-lane swizzle, so the
// fragment addresses of a wave differ by compile-time constants (ds_read offset immediates)
// Measured on gfx950 (tools/probes/lds_bank_probe.hip times all 4096 GF(2)-linear row->chunk maps): with this
// map the fragment ds_read_b128 pattern (16 rows x 4 k-chunks, 128-B rows) issues at the conflict-free
// 4 clk/instruction AND the transposing ds_read_b64_tr_b16 pattern of the attention backward at 2.4 clk (best
// found 2.3).  The textbook (row & 7) XOR costs 7 clk resp. 4 clk: the 64 x 4-B banks serve 16-byte accesses
// in lane groups that are not 16 consecutive lanes, so "8 rows -> 8 chunks" is not enough.
__device__ __forceinline__ int swz(int row) { return ((row ^ (row >> 1)) & 3) | (row & 4); }
// B image of the 256x128 ring kernel: its fragments take tile rows (rho >> 2) * 16 + j * 4 + (rho & 3), so the map
// is applied to the fragment-local row index rho = ((row >> 4) & 3) * 4 + (row & 3) (same lane pattern as above)
__device__ __forceinline__ int swz_b(int row) { return swz((((row >> 4) & 3) << 2) | (row & 3)); }
// The register-staged kernel also WRITES its image with transposing 8-byte stores (K-major operands), whose
// conflicts the map above doubles (TT weight-gradient products 332 -> 389 us); it keeps the textbook map.
__device__ __forceinline__ int swz_rs(int row) { return (row & 7) ^ ((row >> 3) & 1); }

// ------------------------------------------------------------------------------ epilogue (shared)
template <typename TC>
__device__ __forceinline__ void epilogue_store4(const GemmArgs& g, TC* __restrict__ C,
                                                const TC* __restrict__ aux_in, TC* __restrict__ aux_out,
                                                const float* __restrict__ bias, int m, int n0,
                                                const float acc[4], bool lead) {
  if (m >= g.M) return;
  const float rs = (g.epilogue == W2V2_EPI_SCALE_RC) ? g.row_scale[m] : 1.0f;
  float out[4];
  float pre[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j;
    float v = acc[j] * g.alpha;
    pre[j] = 0.f;
    if (n < g.N) {
      switch (g.epilogue) {
        case W2V2_EPI_BIAS:
          if (lead) v += bias[n];
          break;
        case W2V2_EPI_BIAS_GELU:
          v += bias[n];
          pre[j] = v;
          v = gelu_f(v);
          break;
        case W2V2_EPI_BIAS_GELU_GRAD:
          v += bias[n];
          gelu_both_f(v, v, pre[j]);
          break;
        case W2V2_EPI_GELU_BWD:
          v *= gelu_grad_f(to_f32<TC>(aux_in[(int64_t)m * g.ldaux + n]));
          break;
        case W2V2_EPI_MUL:
          v *= to_f32<TC>(aux_in[(int64_t)m * g.ldaux + n]);
          break;
        case W2V2_EPI_ADD:
          v += to_f32<TC>(aux_in[(int64_t)m * g.ldaux + n]);
          break;
        case W2V2_EPI_SCALE_RC:
          v *= rs * g.col_scale[n];
          break;
        default:
          break;
      }
    }
    out[j] = v;
  }
  TC* crow = C + (int64_t)m * g.ldc;
  if (g.atomic) {
    if constexpr (sizeof(TC) == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < g.N) unsafeAtomicAdd(reinterpret_cast<float*>(crow) + n0 + j, out[j]);
    }
    return;
  }
  if (g.c_vec_ok && n0 + 4 <= g.N) {
    if constexpr (sizeof(TC) == 4) {
      *reinterpret_cast<float4*>(crow + n0) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
      uint2 w;
      w.x = pack2<TC>(out[0], out[1]);
      w.y = pack2<TC>(out[2], out[3]);
      *reinterpret_cast<uint2*>(crow + n0) = w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < g.N) crow[n0 + j] = from_f32<TC>(out[j]);
  }
  if ((g.epilogue == W2V2_EPI_BIAS_GELU || g.epilogue == W2V2_EPI_BIAS_GELU_GRAD) && aux_out != nullptr) {
    TC* arow = aux_out + (int64_t)m * g.ldaux;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < g.N) arow[n0 + j] = from_f32<TC>(pre[j]);
  }
}

// ------------------------------------------------------------------------------ coalesced tile epilogue
// The MFMA accumulator layout gives each lane 4 consecutive columns of 16 different rows: storing
// that directly issues 32-byte row fragments.  Instead the tile goes through LDS (the operand tiles
// are dead by now): row-halves of the block tile are written as f32 [rows][BN+4], then all threads
// read back whole rows, apply the epilogue on 8 consecutive columns and issue 16-byte stores,
// 16 lanes per 256-B row segment.
// Epilogue of 8 consecutive columns of one row.  Per-column operands (bias / column scale) are loaded
// ONCE per thread (a thread keeps the same 8 columns for every row it stores) and the aux rows of a
// whole pass are fetched up front with 16-byte loads, so no global-load latency sits between the LDS
// read-back and the store.  The kind is wave-uniform: one scalar branch selects a specialised body.
template <typename TC, int EPI, bool DEFER = false>
__device__ __forceinline__ void epilogue_row8_impl(const GemmArgs& g, TC* __restrict__ Cz, TC* __restrict__ auxz,
                                                   int m, int n, float (&v)[8], const float (&cv)[8],
                                                   const float (&ax)[8], bool lead, uint4* dout = nullptr) {
  const bool full = n + 8 <= g.N;
  float pre[8];
  float rs = 1.0f;
  if constexpr (EPI == W2V2_EPI_SCALE_RC) rs = g.row_scale[m];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = v[e] * g.alpha;
    if constexpr (EPI == W2V2_EPI_BIAS) { if (lead) x += cv[e]; }
    if constexpr (EPI == W2V2_EPI_BIAS_GELU) { x += cv[e]; pre[e] = x; x = gelu_f(x); }
    if constexpr (EPI == W2V2_EPI_BIAS_GELU_GRAD) { x += cv[e]; gelu_both_f(x, x, pre[e]); }
    if constexpr (EPI == W2V2_EPI_GELU_BWD) x *= gelu_grad_f(ax[e]);
    if constexpr (EPI == W2V2_EPI_MUL) x *= ax[e];
    if constexpr (EPI == W2V2_EPI_ADD) x += ax[e];
    if constexpr (EPI == W2V2_EPI_SCALE_RC) x *= rs * cv[e];
    v[e] = x;
  }
  TC* cp = Cz + (int64_t)m * g.ldc + n;
  if constexpr (DEFER) {             // deferred store (host guarantees full, aligned, non-atomic, single output)
    if constexpr (sizeof(TC) == 2)
      *dout = make_uint4(pack2<TC>(v[0], v[1]), pack2<TC>(v[2], v[3]), pack2<TC>(v[4], v[5]), pack2<TC>(v[6], v[7]));
    return;
  }
  if (g.atomic) {
    if constexpr (sizeof(TC) == 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < g.N) unsafeAtomicAdd(reinterpret_cast<float*>(cp) + e, v[e]);
    }
  } else if (full && g.c_vec_ok) {
    Vec8<TC> t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t.v[e] = v[e];
    t.store(cp);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < g.N) cp[e] = from_f32<TC>(v[e]);
  }
  if constexpr (EPI == W2V2_EPI_BIAS_GELU || EPI == W2V2_EPI_BIAS_GELU_GRAD) {
    if (auxz != nullptr) {
      TC* ap = auxz + (int64_t)m * g.ldaux + n;
      if (full && g.aux_vec_ok) {
        Vec8<TC> t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t.v[e] = pre[e];
        t.store(ap);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (n + e < g.N) ap[e] = from_f32<TC>(pre[e]);
      }
    }
  }
}

// read back NIT row-chunks of the staged f32 tile (ROWS x BN, pitch BN+4) and store them
template <typename TC, int EPI, int NIT, int NTHREADS, int BN>
__device__ __forceinline__ void epilogue_pass(const GemmArgs& g, const float* __restrict__ stage,
                                              TC* __restrict__ Cz, TC* __restrict__ auxz, int mbase, int n0,
                                              const float (&cv)[8], bool lead) {
  constexpr int PITCH = BN + 4, CPR = BN / 8;
  const int tid = threadIdx.x;
  const int ch = tid % CPR;                       // the same 8 columns for every iteration
  const int n = n0 + ch * 8;
  if (n >= g.N) return;
  float ax[NIT][8];
  if constexpr (EPI == W2V2_EPI_GELU_BWD || EPI == W2V2_EPI_ADD || EPI == W2V2_EPI_MUL) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = (tid + NTHREADS * it) / CPR;
      const int m = mbase + r;
#pragma unroll
      for (int e = 0; e < 8; ++e) ax[it][e] = 0.f;
      if (m < g.M) {
        const TC* ap = auxz + (int64_t)m * g.ldaux + n;
        if (n + 8 <= g.N && g.aux_vec_ok) {
          Vec8<TC> t;
          t.load(ap);
#pragma unroll
          for (int e = 0; e < 8; ++e) ax[it][e] = t.v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (n + e < g.N) ax[it][e] = to_f32<TC>(ap[e]);
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = (tid + NTHREADS * it) / CPR;
    const int m = mbase + r;
    if (m >= g.M) continue;
    const float4 lo = *reinterpret_cast<const float4*>(stage + r * PITCH + ch * 8);
    const float4 hi = *reinterpret_cast<const float4*>(stage + r * PITCH + ch * 8 + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if constexpr (EPI == W2V2_EPI_GELU_BWD || EPI == W2V2_EPI_ADD || EPI == W2V2_EPI_MUL)
      epilogue_row8_impl<TC, EPI>(g, Cz, auxz, m, n, v, cv, ax[it], lead);
    else
      epilogue_row8_impl<TC, EPI>(g, Cz, auxz, m, n, v, cv, cv, lead);
  }
}

// per-thread column operands: bias[n..n+7] or col_scale[n..n+7]
__device__ __forceinline__ void load_col8(const GemmArgs& g, const float* __restrict__ bias, int n, float (&cv)[8]) {
  const float* src = (g.epilogue == W2V2_EPI_SCALE_RC) ? g.col_scale : bias;
#pragma unroll
  for (int e = 0; e < 8; ++e) cv[e] = 0.f;
  if (src == nullptr || n >= g.N) return;
  if (n + 8 <= g.N && ((reinterpret_cast<uintptr_t>(src + n) & 15) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(src + n), b = *reinterpret_cast<const float4*>(src + n + 4);
    cv[0] = a.x; cv[1] = a.y; cv[2] = a.z; cv[3] = a.w; cv[4] = b.x; cv[5] = b.y; cv[6] = b.z; cv[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < g.N) cv[e] = src[n + e];
  }
}

template <typename TC, int EPI, int FM, int I0, bool DEFER = false, int NR = 4>
__device__ __forceinline__ void epilogue_direct4(const GemmArgs& g, TC* __restrict__ Cz, TC* __restrict__ auxz,
                                                 f32x4 (&acc)[FM][4], int m, int n, const float (&cv0)[8],
                                                 const float (&cv1)[8], uint4 (&pend)[8]) {
  // row fragments I0 .. I0+NR-1 of the wave tile (rows m + 16 i); NR at a time bounds the aux staging registers
  float ax[NR][16];
  if constexpr (EPI == W2V2_EPI_GELU_BWD || EPI == W2V2_EPI_ADD || EPI == W2V2_EPI_MUL) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int mi = m + 16 * (I0 + i);
#pragma unroll
      for (int e = 0; e < 16; ++e) ax[i][e] = 0.f;
      if (mi < g.M) {
        const TC* ap = auxz + (int64_t)mi * g.ldaux + n;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (n + 8 * h + 8 <= g.N && g.aux_vec_ok) {
            Vec8<TC> t;
            t.load(ap + 8 * h);
#pragma unroll
            for (int e = 0; e < 8; ++e) ax[i][8 * h + e] = t.v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + 8 * h + e < g.N) ax[i][8 * h + e] = to_f32<TC>(ap[8 * h + e]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int mi = m + 16 * (I0 + i);
    if (mi >= g.M) continue;                           // (deferred: the flush repeats this test)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (n + 8 * h >= g.N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc[I0 + i][2 * h + (e >> 2)][e & 3];
      float a8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a8[e] = (EPI == W2V2_EPI_GELU_BWD || EPI == W2V2_EPI_ADD || EPI == W2V2_EPI_MUL) ? ax[i][8 * h + e] : 0.f;
      if constexpr (DEFER && FM == 4)
        epilogue_row8_impl<TC, EPI, true>(g, Cz, auxz, mi, n + 8 * h, v, h ? cv1 : cv0, a8, true, &pend[i * 2 + h]);
      else
        epilogue_row8_impl<TC, EPI>(g, Cz, auxz, mi, n + 8 * h, v, h ? cv1 : cv0, a8, true);
    }
  }
}
// epilogue straight from the accumulators of a wave whose lane owns columns n .. n+15 of rows m + 16 i, i < FM.
// NR = row fragments whose aux rows are staged together: 4 (64 VGPRs) hides their load latency best, the kernels at the
// 256-register limit take 2 -- with 4 their tile-loop invariants spilled, and a kernel that touches scratch at all pays
// ~8 us per dispatch (tools/probes/scratch_probe.hip)
template <typename TC, int EPI, int FM, int NR = 4>
__device__ __forceinline__ void epilogue_direct(const GemmArgs& g, TC* __restrict__ Cz, TC* __restrict__ auxz,
                                                f32x4 (&acc)[FM][4], int m, int n, const float (&cv0)[8],
                                                const float (&cv1)[8]) {
  if (n >= g.N) return;
  uint4 unused[8];
  if constexpr (NR == 4) {
    epilogue_direct4<TC, EPI, FM, 0>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
    if constexpr (FM > 4) epilogue_direct4<TC, EPI, FM, 4>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
  } else {
    static_assert(NR == 2 && FM == 8, "two-row batches are wired for the 128-row wave tiles");
    epilogue_direct4<TC, EPI, FM, 0, false, 2>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
    epilogue_direct4<TC, EPI, FM, 2, false, 2>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
    epilogue_direct4<TC, EPI, FM, 4, false, 2>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
    epilogue_direct4<TC, EPI, FM, 6, false, 2>(g, Cz, auxz, acc, m, n, cv0, cv1, unused);
  }
}

// ------------------------------------------------------------------------------ full-line register epilogue
// In the layout above a lane (c = lane & 15, q = lane >> 4) holds, for row fragment i, the 16 columns nc .. nc + 15 of
// row 16 i + c as two 16-byte halves P0 | P1.  Stored directly, one wave instruction writes, per row, FOUR SCATTERED
// 16-byte pieces (q * 32 bytes apart): 4.6 TB/s over the whole chip (tools/probes/store_pattern_probe), 4.9 with 64-byte
// runs, 5.8 when 8 lanes cover one whole 128-byte line.  So lanes c and c ^ 8 swap one half each (v_mov_dpp row_ror:8:
// four moves per fragment and output plane): lanes c < 8 keep P0 and receive the partner's P0, lanes c >= 8 keep P1 and
// receive the partner's P1 -- every lane then stores columns nc + (c < 8 ? 0 : 8) .. + 7 of rows 16 i + (c & 7) and + 8,
// and one instruction writes 8 rows x 128 contiguous bytes.  aux rows are FETCHED in the same pattern (packed, four
// fragments = 32 registers at a time, all loads of a batch in flight before the first use) and swapped back.
// host-side eligibility (wave-uniform): 16-bit C (and aux), 16-byte aligned rows, every 64-column wave tile inside N
__device__ __forceinline__ bool lines_ok(const GemmArgs& g) {
  return g.c_vec_ok && !g.atomic && (g.N & 63) == 0 && (g.aux == nullptr || g.aux_vec_ok);
}

// mw = first row of the wave tile (no lane part), nc = this lane's first column; rows mw + 16 i + c, i < FM.
// DEFER: the two 16-byte stores of fragment i are left in pend[2 i], pend[2 i + 1] (see lines_flush).
template <typename TC, int EPI, int FM, bool DEFER = false>
__device__ __forceinline__ void epilogue_lines(const GemmArgs& g, TC* __restrict__ Cz, TC* __restrict__ auxz,
                                               f32x4 (&acc)[FM][4], int mw, int nc, int lane, const float (&cv0)[8],
                                               const float (&cv1)[8], uint4* __restrict__ pend) {
  static_assert(sizeof(TC) == 2, "16-bit outputs only");
  constexpr bool READS_AUX = EPI == W2V2_EPI_GELU_BWD || EPI == W2V2_EPI_ADD || EPI == W2V2_EPI_MUL;
  constexpr bool WRITES_AUX = EPI == W2V2_EPI_BIAS_GELU || EPI == W2V2_EPI_BIAS_GELU_GRAD;
  if constexpr (!DEFER) {
    if (nc >= g.N) return;                       // wave tile wholly past the last column (uniform: N % 64 == 0)
  }
  const bool wt = g.wt_stores != 0;               // (wave-uniform: one scalar branch per store)
  auto st16 = [wt](TC* p, uint4 v) {
    if (wt) store16_wt(p, v); else *reinterpret_cast<uint4*>(p) = v;
  };
  const int c = lane & 15;
  const bool lo = c < 8;
  const int ra = mw + (c & 7);                   // rows this lane stores / fetches: ra + 16 i, ra + 16 i + 8
  const int ncs = nc + (lo ? 0 : 8);
  const int mo = mw + c;                         // row of this lane's own values: mo + 16 i
#pragma unroll
  for (int i0 = 0; i0 < FM; i0 += 4) {
    uint4 la[4], lb[4];
    if constexpr (READS_AUX) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = ra + 16 * (i0 + i);
        la[i] = lb[i] = make_uint4(0, 0, 0, 0);
        if (r < g.M) la[i] = *reinterpret_cast<const uint4*>(auxz + (int64_t)r * g.ldaux + ncs);
        if (r + 8 < g.M) lb[i] = *reinterpret_cast<const uint4*>(auxz + (int64_t)(r + 8) * g.ldaux + ncs);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float ax[16];
      if constexpr (READS_AUX) {
        uint4 a0, a1;
        lines_to_halves(lo, la[i], lb[i], a0, a1);
        unpack8<TC>(a0, ax);
        unpack8<TC>(a1, ax + 8);
      }
      float rs = 1.0f;
      if constexpr (EPI == W2V2_EPI_SCALE_RC) rs = (mo + 16 * (i0 + i) < g.M) ? g.row_scale[mo + 16 * (i0 + i)] : 0.f;
      float v[16], pre[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float x = acc[i0 + i][e >> 2][e & 3] * g.alpha;
        const float cve = e < 8 ? cv0[e] : cv1[e - 8];
        if constexpr (EPI == W2V2_EPI_BIAS) x += cve;
        if constexpr (EPI == W2V2_EPI_BIAS_GELU) { x += cve; pre[e] = x; x = gelu_f(x); }
        if constexpr (EPI == W2V2_EPI_BIAS_GELU_GRAD) { x += cve; gelu_both_f(x, x, pre[e]); }
        if constexpr (EPI == W2V2_EPI_GELU_BWD) x *= gelu_grad_f(ax[e]);
        if constexpr (EPI == W2V2_EPI_MUL) x *= ax[e];
        if constexpr (EPI == W2V2_EPI_ADD) x += ax[e];
        if constexpr (EPI == W2V2_EPI_SCALE_RC) x *= rs * cve;
        v[e] = x;
      }
      uint4 da, db;
      halves_to_lines(lo, pack8<TC>(v), pack8<TC>(v + 8), da, db);
      const int r = ra + 16 * (i0 + i);
      if constexpr (DEFER) {
        pend[2 * (i0 + i)] = da;
        pend[2 * (i0 + i) + 1] = db;
      } else {
        if (r < g.M) st16(Cz + (int64_t)r * g.ldc + ncs, da);
        if (r + 8 < g.M) st16(Cz + (int64_t)(r + 8) * g.ldc + ncs, db);
      }
      if constexpr (WRITES_AUX) {
        if (auxz != nullptr) {
          uint4 xa, xb;
          halves_to_lines(lo, pack8<TC>(pre), pack8<TC>(pre + 8), xa, xb);
          if (r < g.M) st16(auxz + (int64_t)r * g.ldaux + ncs, xa);
          if (r + 8 < g.M) st16(auxz + (int64_t)(r + 8) * g.ldaux + ncs, xb);
        }
      }
    }
  }
}

#define W2V2_EPI_DISPATCH(CALL)                                              \
  switch (g.epilogue) {                                                      \
    case W2V2_EPI_BIAS: { constexpr int EPI = W2V2_EPI_BIAS; CALL; } break;  \
    case W2V2_EPI_BIAS_GELU: { constexpr int EPI = W2V2_EPI_BIAS_GELU; CALL; } break; \
    case W2V2_EPI_GELU_BWD: { constexpr int EPI = W2V2_EPI_GELU_BWD; CALL; } break;   \
    case W2V2_EPI_ADD: { constexpr int EPI = W2V2_EPI_ADD; CALL; } break;    \
    case W2V2_EPI_SCALE_RC: { constexpr int EPI = W2V2_EPI_SCALE_RC; CALL; } break;   \
    case W2V2_EPI_BIAS_GELU_GRAD: { constexpr int EPI = W2V2_EPI_BIAS_GELU_GRAD; CALL; } break; \
    case W2V2_EPI_MUL: { constexpr int EPI = W2V2_EPI_MUL; CALL; } break;    \
    default: { constexpr int EPI = W2V2_EPI_NONE; CALL; } break;             \
  }

// ------------------------------------------------------------------------------ coalesced tile epilogue (128-row tiles)
template <typename TC, int FM, int FN>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, f32x4 (&acc)[FM][FN], float* __restrict__ stage,
                                              int m0, int n0, int wm, int wn, int z0, int z1, int split) {
  constexpr int BN = 32 * FN, ROWS = 16 * FM, PITCH = BN + 4, CPR = BN / 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int frow = lane & 15, fk = lane >> 4;
  TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  float cv[8];
  load_col8(g, bias, n0 + (tid % CPR) * 8, cv);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<float4*>(stage + (i * 16 + frow) * PITCH + wn * (16 * FN) + j * 16 + fk * 4) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
    W2V2_EPI_DISPATCH((epilogue_pass<TC, EPI, (ROWS * CPR) / 256, 256, BN>(g, stage, Cz, auxz, m0 + pass * ROWS, n0, cv,
                                                                        split == 0)));
  }
}

// Per-lane global source pointers of NP DMA pieces of one operand tile: piece j covers tile row row0 + j * step (this
// lane's row of the piece), at element column col[j] of the K-contiguous row.  The general form costs a clamp, a
// (segmented: 64-bit division) row offset and a 64-bit multiply PER PIECE -- with six pieces ~600 serially dependent
// instructions = 1.7 us between kernel entry and the first DMA (s_memtime stamps), paid again for every tile of a
// persistent workgroup.  When the whole tile lies inside the operand and inside ONE segment (uniform test, scalar
// unit) the pieces are an arithmetic progression: one row offset, then NP - 1 additions.
template <int NP>
__device__ __forceinline__ void tile_ptrs(const OpDev& o, const bf16_t* base, int t0, int TR, int bound, int row0,
                                          int step, const int* col, const bf16_t** out) {
  bool fast = t0 + TR <= bound;
  int64_t seg_base = 0;
  int first = t0;
  if (o.seg_len > 0) {
    const int sl = (int)o.seg_len;
    const int q0 = t0 / sl, q1 = (t0 + TR - 1) / sl;       // uniform 32-bit divisions (row counts fit an int)
    fast = fast && q0 == q1;
    seg_base = (int64_t)q0 * o.seg_stride;
    first = t0 - q0 * sl;
  }
  if (fast) {
    const bf16_t* p0 = base + seg_base + (int64_t)(first + row0) * o.ld;
#pragma unroll
    for (int j = 0; j < NP; ++j) out[j] = p0 + (int64_t)(j * step) * o.ld + col[j];
  } else {
#pragma unroll
    for (int j = 0; j < NP; ++j) out[j] = base + outer_off(o, min(t0 + row0 + j * step, bound - 1)) + col[j];
  }
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); remap so each
// XCD owns a contiguous run of tiles (neighbouring tiles share the A row panel in its private L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}


typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// CUs the persistent grids may fill (W2V2_RESERVE_CUS keeps some out for RCCL's channels); gemm.hip
// Kernel-timestamp timing of single launches (w2v2_gemm_timed): when a slot's events are pending, the launch helpers of
// the two dominant kernels hand them to hipExtLaunchKernelGGL, which stamps the dispatch's own begin / end (what
// rocprofv3 reports) instead of bracketing the launch with two stream events (+3 us of dispatch time per launch).
struct W2v2PendingTimer { hipEvent_t start, stop; bool armed; };
W2v2PendingTimer& w2v2_pending_timer();
#define W2V2_LAUNCH_MAYBE_TIMED(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS)                                         \
  do {                                                                                                          \
    W2v2PendingTimer& pt_ = w2v2_pending_timer();                                                               \
    if (pt_.armed) {                                                                                            \
      pt_.armed = false;                                                                                        \
      hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, pt_.start, pt_.stop, 0, ARGS);                    \
    } else {                                                                                                    \
      hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, ARGS);                                               \
    }                                                                                                           \
  } while (0)
int w2v2_gemm_device_cus();
// cross-file launchers: dtype_ab / dtype_c are the W2V2_* codes (16-bit operands; C 16-bit of the same type or f32)
void w2v2_launch_ring_256x128(const GemmArgs& a, int dtype_ab, int dtype_c, int M, int N, int batch, bool persistent,
                              hipStream_t st);
void w2v2_launch_phased_256x256(const GemmArgs& a, int dtype_ab, int dtype_c, int M, int N, int batch, hipStream_t st);
// gemm_f32.hip: exact-f32 products (f32 operands, f32 C); split = split-K factor (atomics), chooses its own tile
void w2v2_launch_gemm_f32(GemmArgs a, int M, int N, int K, int split, int batch, hipStream_t st);
int w2v2_gemm_f32_dma_rows(const GemmArgs& a, int M, int N, int K, int split, int batch);   // gemm_f32.hip: 0 = register-staged
// gemm_f32_dma.hip: the LDS-DMA variant, (32 fi) x 128 tiles, nst-stage ring; the caller has checked eligibility
void w2v2_launch_gemm_f32_dma(const GemmArgs& a, int M, int N, int split, int batch, int fi, int nst, hipStream_t st);
