// gemm.hip -- the one matmul-shaped kernel of the path (see include/w2v2_hip.h, "GEMM").
//
//   bf16 path : 128x128 (or 128x64) x 64 block tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16,
//               f32 accumulate.  Operands are staged global -> VGPR -> LDS (one barrier per K tile,
//               next tile's global loads in flight under the MFMAs).  K-major ("trans") operands
//               are transposed in registers (4k x 8m micro-blocks) on their way to LDS, so the
//               forward (NT), data-gradient (NN) and weight-gradient (TN) products all run on the
//               same kernel with no transposed copies in HBM.  The segmented outer index gives
//               zero-copy implicit im2col for the strided conv stack and the grouped pos-conv.
//               LDS image: [row][64 k] bf16, 128-B rows, 16-B chunks XOR-swizzled by
//               (row ^ row>>3) & 7 -> conflict-free ds_read_b128 fragment reads and ds_write_b128
//               staging writes, <= 2-way on the transposing ds_write_b64.
//   f32 path  : exact-f32 64x64x16 VALU tile kernel.  It exists for the parity mode only
//               (embeddings within 1e-3 rel-L2 of the f32 reference, BASELINE.json north_star);
//               throughput runs use bf16.
#include "common.h"
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
  // two-term weights (w2v2_hip.h): tiles with n0 >= n_ext_from run k_ext more K steps against B + b_lo_off
  int k_ext, n_ext_from;
  int64_t b_lo_off;
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
        if (r < g.M) store16_wt(Cz + (int64_t)r * g.ldc + ncs, da);
        if (r + 8 < g.M) store16_wt(Cz + (int64_t)(r + 8) * g.ldc + ncs, db);
      }
      if constexpr (WRITES_AUX) {
        if (auxz != nullptr) {
          uint4 xa, xb;
          halves_to_lines(lo, pack8<TC>(pre), pack8<TC>(pre + 8), xa, xb);
          if (r < g.M) store16_wt(auxz + (int64_t)r * g.ldaux + ncs, xa);
          if (r + 8 < g.M) store16_wt(auxz + (int64_t)(r + 8) * g.ldaux + ncs, xb);
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

// ------------------------------------------------------------------------------ bf16 MFMA kernel
// stage one operand tile (R outer rows x 64 k) global -> registers
template <int R, bool TRANS>
struct Stager {
  static constexpr int NV = TRANS ? 4 : (R / 32);  // uint4 per thread
  uint4 v[NV];

  __device__ __forceinline__ void load(const OpDev& o, const bf16_t* __restrict__ base, int r0,
                                       int rbound, int k0, int kend, int tid,
                                       const int64_t* __restrict__ rowoff) {
    if constexpr (!TRANS) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = tid + 256 * j;
        const int row = c >> 3, kc = c & 7;
        const int k = k0 + kc * 8;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (r0 + row < rbound && k < kend) {
          const bf16_t* p = base + rowoff[j] + k;
          if (o.vec_ok && k + 8 <= kend) {
            val = *reinterpret_cast<const uint4*>(p);
          } else {
            bf16_t t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (k + e < kend) ? p[e] : (bf16_t)0;
            val.x = t[0] | ((uint32_t)t[1] << 16); val.y = t[2] | ((uint32_t)t[3] << 16);
            val.z = t[4] | ((uint32_t)t[5] << 16); val.w = t[6] | ((uint32_t)t[7] << 16);
          }
        }
        v[j] = val;
      }
    } else {
      // 4 k-rows x 8 inner per thread; block id = tid (+ nothing: R*2 blocks, R in {64,128})
      const int nblk = R * 2;
      const int mb = tid % (R / 8), kb = tid / (R / 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 val = make_uint4(0, 0, 0, 0);
        const int k = k0 + kb * 4 + i;
        const int in0 = r0 + mb * 8;
        if (tid < nblk && k < kend && in0 < rbound) {
          const bf16_t* p = base + outer_off(o, k) + in0;
          if (o.vec_ok && in0 + 8 <= rbound) {
            val = *reinterpret_cast<const uint4*>(p);
          } else {
            bf16_t t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (in0 + e < rbound) ? p[e] : (bf16_t)0;
            val.x = t[0] | ((uint32_t)t[1] << 16); val.y = t[2] | ((uint32_t)t[3] << 16);
            val.z = t[4] | ((uint32_t)t[5] << 16); val.w = t[6] | ((uint32_t)t[7] << 16);
          }
        }
        v[i] = val;
      }
    }
  }

  __device__ __forceinline__ void store(bf16_t* __restrict__ lds, int tid) const {
    if constexpr (!TRANS) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = tid + 256 * j;
        const int row = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(lds + row * 64 + ((kc ^ swz_rs(row)) << 3)) = v[j];
      }
    } else {
      const int nblk = R * 2;
      if (tid >= nblk) return;
      const int mb = tid % (R / 8), kb = tid / (R / 8);
      const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w},
                                {v[1].x, v[1].y, v[1].z, v[1].w},
                                {v[2].x, v[2].y, v[2].z, v[2].w},
                                {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int d = mi >> 1;
        uint2 o2;
        if (mi & 1) {
          o2.x = (w[0][d] >> 16) | (w[1][d] & 0xffff0000u);
          o2.y = (w[2][d] >> 16) | (w[3][d] & 0xffff0000u);
        } else {
          o2.x = (w[0][d] & 0xffffu) | (w[1][d] << 16);
          o2.y = (w[2][d] & 0xffffu) | (w[3][d] << 16);
        }
        const int row = mb * 8 + mi;
        const int chunk = kb >> 1, half = kb & 1;
        *reinterpret_cast<uint2*>(lds + row * 64 + ((chunk ^ swz_rs(row)) << 3) + half * 4) = o2;
      }
    }
  }
};

template <typename TE, int FM, int FN, bool TA, bool TB, typename TC>
__global__ __launch_bounds__(256) void gemm16_regstage_kernel(const GemmArgs g) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);           // [2][BM*64]
  bf16_t* Bs = As + 2 * BM * 64;                              // [2][BN*64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntile = g.tiles_m * g.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int split = blockIdx.y;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  // row offsets of the non-transposed operands are K-invariant: compute once
  int64_t arow[TA ? 1 : (BM / 32)], brow[TB ? 1 : (BN / 32)];
  if constexpr (!TA) {
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) {
      const int row = (tid + 256 * j) >> 3;
      arow[j] = (m0 + row < g.M) ? outer_off(g.A, m0 + row) : 0;
    }
  }
  if constexpr (!TB) {
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) {
      const int row = (tid + 256 * j) >> 3;
      brow[j] = (n0 + row < g.N) ? outer_off(g.B, n0 + row) : 0;
    }
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager<BM, TA> sa;
  Stager<BN, TB> sb;
  const int nk = (kend - kbeg + 63) / 64;
  if (nk > 0) {
    sa.load(g.A, Ab, m0, g.M, kbeg, kend, tid, arow);
    sb.load(g.B, Bb, n0, g.N, kbeg, kend, tid, brow);
    sa.store(As, tid);
    sb.store(Bs, tid);
  }
  __syncthreads();

  const int frow = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k0 = kbeg + (kt + 1) * 64;
      sa.load(g.A, Ab, m0, g.M, k0, kend, tid, arow);
      sb.load(g.B, Bb, n0, g.N, k0, kend, tid, brow);
    }
    const bf16_t* Ac = As + cur * BM * 64;
    const bf16_t* Bc = Bs + cur * BN * 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * (16 * FM) + i * 16 + frow;
        af[i] = *reinterpret_cast<const frag8_t*>(Ac + row * 64 + (((kk * 4 + fk) ^ swz_rs(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * (16 * FN) + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const frag8_t*>(Bc + row * 64 + (((kk * 4 + fk) ^ swz_rs(row)) << 3));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // operands swapped: D[row = n][col = m] so each lane owns 4 consecutive n of one m
          acc[i][j] = mfma16<TE>(bfr[j], af[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      sa.store(As + (cur ^ 1) * BM * 64, tid);
      sb.store(Bs + (cur ^ 1) * BN * 64, tid);
    }
    __syncthreads();
  }

  tile_epilogue<TC, FM, FN>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, z0, z1, split);
}

// ------------------------------------------------------------------------------ bf16 MFMA kernel, LDS-DMA staging
// Fast path for K-contiguous operands (forward products and, with the pre-transposed weight copies,
// the data-gradient products): tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR
// round trip, no ds_write pass -- the staging writes were the LDS bottleneck of the register-staged
// kernel).  An LDS-DMA wave-instruction writes 1 KiB lane-linearly (8 rows x 128 B), so the XOR
// swizzle is applied on the per-lane SOURCE address (physical chunk c' of row r loads logical chunk
// c' ^ swz(r)) and again on the fragment read.  Out-of-range rows are clamped to the last valid row
// (their results are never stored); K must be a multiple of 64.
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <typename TE, int FM, int FN, typename TC>
__global__ __launch_bounds__(256) void gemm16_dma_128_kernel(const GemmArgs g) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  constexpr int NA = BM / 32, NB = BN / 32;      // 1 KiB pieces per wave per operand tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);           // [2][BM*64]
  bf16_t* Bs = As + 2 * BM * 64;                              // [2][BN*64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntile = g.tiles_m * g.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int split = blockIdx.y;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg) >> 6;

  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  const int c8 = lane & 7, r8 = lane >> 3;
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int row = (wave * NA + j) * 8 + r8;
    const int grow = min(m0 + row, g.M - 1);
    ap[j] = Ab + outer_off(g.A, grow) + kbeg + ((c8 ^ swz(row)) << 3);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = (wave * NB + j) * 8 + r8;
    const int grow = min(n0 + row, g.N - 1);
    bp[j] = Bb + outer_off(g.B, grow) + kbeg + ((c8 ^ swz(row)) << 3);
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int buf, int kt) {
    bf16_t* ad = As + buf * BM * 64 + wave * NA * 8 * 64;
    bf16_t* bd = Bs + buf * BN * 64 + wave * NB * 8 * 64;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * 64), (lvoid_t*)(ad + j * 8 * 64), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kt * 64), (lvoid_t*)(bd + j * 8 * 64), 16, 0, 0);
  };
  const int frow = lane & 15, fk = lane >> 4;
  auto compute = [&](int buf) {
    const bf16_t* Ac = As + buf * BM * 64;
    const bf16_t* Bc = Bs + buf * BN * 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * (16 * FM) + i * 16 + frow;
        af[i] = *reinterpret_cast<const frag8_t*>(Ac + row * 64 + (((kk * 4 + fk) ^ swz(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * (16 * FN) + j * 16 + frow;
        bfr[j] = *reinterpret_cast<const frag8_t*>(Bc + row * 64 + (((kk * 4 + fk) ^ swz(row)) << 3));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<TE>(bfr[j], af[i], acc[i][j]);
    }
  };

  if (nk > 0) stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) stage(1, kt + 1);
    compute(0);
    __syncthreads();
    if (kt + 1 < nk) {
      if (kt + 2 < nk) stage(0, kt + 2);
      compute(1);
      __syncthreads();
    }
  }

  TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const TC* auxz = g.aux ? reinterpret_cast<const TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  TC* auxo = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  tile_epilogue<TC, FM, FN>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, z0, z1, split);
}

// ------------------------------------------------------------------------------ 256x128x64, 3-stage LDS-DMA ring
// The 128x128 kernel moves 32 KiB from L2 per 2.1 MFLOP (64 FLOP/B): at 2 workgroups per CU that is a
// large fraction of the aggregate L2 bandwidth, and its 1-tile prefetch distance (vmcnt(0) before every
// barrier) exposes the L2/HBM latency once per K tile.  This variant uses a 256x128 block tile (8 waves
// as 4x2, 64x64 per wave, 87 FLOP/B) and a 3-stage LDS ring (144 KiB) with a COUNTED wait: while tile t
// is multiplied, tiles t+1 and t+2 are in flight; per K tile one raw s_barrier and `s_waitcnt vmcnt(G)`
// (G = this wave's DMA pieces per stage), never vmcnt(0) in the loop.
template <int S> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (S == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (S == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (S == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (S == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(S == 0 || S == 4 || S == 6 || S == 8, "unsupported count");
}

template <typename TE, typename TC>
__global__ __launch_bounds__(512) void gemm16_ring_256x128_kernel(const GemmArgs g) {
  constexpr int BM = 256, BN = 128, FM = 4, FN = 4;
  constexpr int STAGE = (BM + BN) * 64;           // elements per stage (A then B)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 1, wn = wave & 1;

  // Deferred stores (g.defer_ok): the 8 x 16-byte stores of a tile are kept packed in registers and issued in the
  // first three ring steps of the NEXT tile (ahead of that step's DMA pieces, so the counted vmcnt waits stay valid):
  // issued together at the tile end they leave at the HBM write rate (~9 B/clk per CU) while nothing else runs.
  uint4 pend[8];
  bool pending = false;
  int pend_m = 0, pend_n = 0;
  TC* const Cdef = reinterpret_cast<TC*>(g.C) + (blockIdx.z / g.batch_inner) * g.c_s0 +
                   (blockIdx.z % g.batch_inner) * g.c_s1;
  // pend[2 i], pend[2 i + 1] = rows pend_m + 16 i and + 8, columns pend_n .. pend_n + 7 (epilogue_lines)
  auto flush = [&](auto first, auto count) {
    if constexpr (sizeof(TC) == 2) {
#pragma unroll
      for (int q = decltype(first)::value; q < decltype(first)::value + decltype(count)::value; ++q) {
        const int mi = pend_m + 16 * (q >> 1) + 8 * (q & 1);
        if (mi < g.M)
          store16_wt(Cdef + (int64_t)mi * g.ldc + pend_n, pend[q]);
      }
    }
  };
  // Persistent over tiles: gridDim.x = min(tiles, CUs) workgroups, each takes tiles t, t + G, ...  (one 144 KiB
  // workgroup per CU anyway).  The next tile's first two DMA stages are issued right behind the epilogue's stores,
  // so their latency -- and a workgroup launch -- hides under the store drain instead of following it.
  const int ntile = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int z = blockIdx.z;
#pragma unroll 1
  for (int t0 = 0; t0 < ntile; t0 += G) {
  const int nchunk = min(G, ntile - t0);
  if ((int)blockIdx.x >= nchunk) break;
  const int tile = t0 + xcd_remap(blockIdx.x, nchunk);
  int tm, tn;
  {
    // Longest tiles first: with two-term weight columns (n >= n_ext_from) a tile of those columns runs twice the K
    // steps.  In plain row-major order a workgroup of the fused QKV product (702 tiles, 3 rounds) can draw two double
    // tiles and a single one (60 K steps against a mean of 44); enumerating all double tiles before the single ones
    // bounds it at 48.  Tiles of one row panel stay adjacent inside each class (L2 reuse of the A rows).
    const int tl = (g.k_ext > 0 && g.n_ext_from > 0) ? min(g.tiles_n, g.n_ext_from / BN) : 0;   // single-K columns
    const int th = g.tiles_n - tl;
    if (tl == 0 || th == 0) {
      tm = tile / g.tiles_n;
      tn = tile - tm * g.tiles_n;
    } else if (tile < g.tiles_m * th) {
      tm = tile / th;
      tn = tl + (tile - tm * th);
    } else {
      const int t2 = tile - g.tiles_m * th;
      tm = t2 / tl;
      tn = t2 - tm * tl;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  // K steps of this tile: nk1 over (A, B) + for the tiles of the two-term weight columns nk - nk1 more over (A, B_lo)
  const int nk1 = g.K >> 6;
  const int nk = nk1 + ((g.k_ext > 0 && n0 >= g.n_ext_from) ? (g.k_ext >> 6) : 0);
  auto koff_a = [&](int kt) -> int { return (kt < nk1 ? kt : kt - nk1) * 64; };
  auto koff_b = [&](int kt) -> int64_t { return kt < nk1 ? (int64_t)kt * 64 : (int64_t)(kt - nk1) * 64 + g.b_lo_off; };

  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  const int c8 = lane & 7, r8 = lane >> 3;
  const bf16_t* ap[4];
  const bf16_t* bp[2];
  {
    int ca[4], cb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) ca[j] = (c8 ^ swz((wave * 4 + j) * 8 + r8)) << 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) cb[j] = (c8 ^ swz_b((wave * 2 + j) * 8 + r8)) << 3;
    tile_ptrs<4>(g.A, Ab, m0, BM, g.M, wave * 32 + r8, 8, ca, ap);
    tile_ptrs<2>(g.B, Bb, n0, BN, g.N, wave * 16 + r8, 8, cb, bp);
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](bf16_t* base, int kt) {
    bf16_t* ad = base + wave * 4 * 8 * 64;
    bf16_t* bd = base + BM * 64 + wave * 2 * 8 * 64;
    const int ka = koff_a(kt);
    const int64_t kb = koff_b(kt);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + ka), (lvoid_t*)(ad + j * 8 * 64), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kb), (lvoid_t*)(bd + j * 8 * 64), 16, 0, 0);
  };
  const int frow = lane & 15, fk = lane >> 4;
  // per-lane fragment offsets (elements) for k-step 0 / 1; everything else is a compile-time constant
  const int lo0 = frow * 64 + ((fk ^ swz(frow)) << 3);
  const int lo1 = frow * 64 + (((4 + fk) ^ swz(frow)) << 3);
  // B fragment j, operand row rho = frow  <->  tile row (rho >> 2) * 16 + j * 4 + (rho & 3): after the MFMA a lane
  // owns the 16 CONSECUTIVE columns fk * 16 + j * 4 + e of its row (register epilogue below)
  const int brow = (frow >> 2) * 16 + (frow & 3);
  const int lb0 = brow * 64 + ((fk ^ swz(frow)) << 3);
  const int lb1 = brow * 64 + (((4 + fk) ^ swz(frow)) << 3);
  const int aoff = wm * 64 * 64, boff = BM * 64 + wn * 64 * 64;
  // piece p of the 6 DMA pieces of one stage: A0..A3, B0, B1
  auto stage_piece = [&](bf16_t* base, int kt, int p) {
    if (p < 4)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[p] + koff_a(kt)), (lvoid_t*)(base + (wave * 4 + p) * 8 * 64), 16,
                                       0, 0);
    else
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[p - 4] + koff_b(kt)),
                                       (lvoid_t*)(base + BM * 64 + (wave * 2 + p - 4) * 8 * 64), 16, 0, 0);
  };
  // multiply stage `base`; when kload >= 0 the DMA pieces of K tile kload go to `nxt`, spread over the MFMA groups
  // (issued back to back behind the barrier they keep both waves of a SIMD in the queue-limited DMA issue)
  auto compute = [&](const bf16_t* base, bf16_t* nxt, int kload) {
    const bf16_t* a0 = base + aoff + lo0;
    const bf16_t* a1 = base + aoff + lo1;
    const bf16_t* b0 = base + boff + lb0;
    const bf16_t* b1 = base + boff + lb1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const frag8_t*>((kk ? a1 : a0) + i * 16 * 64);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const frag8_t*>((kk ? b1 : b0) + j * 4 * 64);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<TE>(bfr[j], af[i], acc[i][j]);
        if (kload >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 0) {
            stage_piece(nxt, kload, i);                  // pieces 0..3 behind the four groups of the first half
          } else if (i < 2) {
            stage_piece(nxt, kload, 4 + i);              // pieces 4, 5
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  bf16_t* s0 = smem;
  bf16_t* s1 = smem + STAGE;
  bf16_t* s2 = smem + 2 * STAGE;

  // one ring step: tile kt is in `cur`; tile kt+2 goes to `nxt` (which held tile kt-1)
#define W2V2_RING_STEP(cur, nxt)                                   \
  {                                                                \
    if (kt + 1 < nk) wait_vmcnt<6>(); else wait_vmcnt<0>();        \
    __builtin_amdgcn_s_barrier();                                  \
    compute(cur, nxt, kt + 2 < nk ? kt + 2 : -1);                  \
    ++kt;                                                          \
  }
  __builtin_amdgcn_s_barrier();          // every wave has finished reading the previous tile's stages
  if (nk > 0) stage(s0, 0);
  if (nk > 1) stage(s1, 1);
  int kt = 0;
  if (pending) {
    if (nk >= 3) {                                         // one peeled rotation of the ring carries the stores
      wait_vmcnt<6>();
      __builtin_amdgcn_s_barrier();
      flush(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
      compute(s0, s2, nk > 2 ? 2 : -1);
      kt = 1;
      wait_vmcnt<6>();
      __builtin_amdgcn_s_barrier();
      flush(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
      compute(s1, s0, nk > 3 ? 3 : -1);
      kt = 2;
      if (nk > 3) wait_vmcnt<6>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      flush(std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
      compute(s2, s1, nk > 4 ? 4 : -1);
      kt = 3;
    } else {
      flush(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    }
    pending = false;
  }
  while (kt < nk) {
    W2V2_RING_STEP(s0, s2)
    if (kt >= nk) break;
    W2V2_RING_STEP(s1, s0)
    if (kt >= nk) break;
    W2V2_RING_STEP(s2, s1)
  }
#undef W2V2_RING_STEP

  // Register epilogue: thanks to the permuted B rows a lane holds, for each of its four rows, 16 consecutive
  // output columns (32 B of bf16): bias / GELU / residual are applied in registers and stored as 2 x 16 B per lane,
  // four lanes covering 128 contiguous bytes of a row -- no LDS round trip and no barrier after the main loop.
  TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  const int nc = n0 + wn * 64 + fk * 16;
  float cv0[8], cv1[8];
  load_col8(g, bias, nc, cv0);
  load_col8(g, bias, nc + 8, cv1);
  if constexpr (sizeof(TC) == 2) {
    if (g.defer_ok) {            // (the host grants defer_ok only where lines_ok holds)
      W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, 4, true>(g, Cz, auxz, acc, m0 + wm * 64, nc, lane, cv0, cv1, pend)));
      pending = true;
      pend_m = m0 + wm * 64 + (frow & 7);
      pend_n = nc + (frow < 8 ? 0 : 8);
    } else if (lines_ok(g)) {
      W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, 4>(g, Cz, auxz, acc, m0 + wm * 64, nc, lane, cv0, cv1, pend)));
    } else {
      W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, 4>(g, Cz, auxz, acc, m0 + wm * 64 + frow, nc, cv0, cv1)));
    }
  } else {
    W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, 4>(g, Cz, auxz, acc, m0 + wm * 64 + frow, nc, cv0, cv1)));
  }
  }   // tile loop
  if (pending) flush(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
}

// ------------------------------------------------------------------------------ 256 x 256 x 32, 4-stage ring
// The 256x128 kernel above is bound by the L2 -> LDS feed (48 KiB per 4.2 MFLOP tile step; all 256 CUs together draw
// ~22 TB/s, tools/probes/load_path_probe).  A 256x256 tile halves the bytes per flop (32 KiB per 4.2 MFLOP step at
// BK = 32).  8 waves as 2 (m) x 4 (n), 128 x 64 per wave (128 accumulator VGPRs); LDS holds FOUR 32 KiB stages
// [256 + 256 rows][32 k] with 64-byte rows, counted `s_waitcnt vmcnt(4)` (4 DMA pieces per wave and stage) and one
// s_barrier per step of 32 MFMAs.  64-byte rows: chunk map s(row) = ((row >> 2) & 1) << 1 is the
// conflict-free one for the fragment ds_read_b128 (lds_bank_probe: 4.0 vs 6.0 clk unswizzled).  Used for products
// whose 256x256 tiling fills the chip (FFN1, dH, the conv stack); B rows permuted for the register epilogue as above.
__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }
__device__ __forceinline__ int swz64_b(int row) { return ((row >> 4) & 1) << 1; }   // fragment-local row (row>>4&3)*4+(row&3)

// Software pipelining of the fragment reads.  Written as "read 12 fragments, then 32 MFMAs" the compiler emits
// `ds_read x6 ; s_waitcnt lgkmcnt(0) ; mfma x8 ; ds_read x2 ; lgkmcnt(0) ; ...`:
// every group of MFMAs waits for a full LDS drain, and since the per-step barrier keeps the two waves of a SIMD in
// phase the matrix pipe idles for each of them (timing with DMA and epilogue switched off: the fragment-read + MFMA
// loop alone runs at ~45 % of the MFMA rate).  Here the fragments of step kt+1 are read WHILE step kt multiplies:
// B fragments double-buffered (16 VGPRs), A fragment i re-loaded in place right after its four MFMAs, so each
// ds_read has ~a full step of MFMA time to land (glds4 590 vs 559 TFLOP/s in-step).  Stage kt+1 must have landed one
// step earlier than in a read-then-multiply loop, i.e. two stages are in flight instead of three; a fifth stage
// (160 KiB, the whole LDS) restored the distance and measured the same, so four it is.
template <typename TE, typename TC>
__global__ __launch_bounds__(512) void gemm16_ring_256x256_kernel(const GemmArgs g) {
  constexpr int BM = 256, BN = 256, FM = 8, FN = 4, BK = 32, S = 4;
  constexpr int STAGE = (BM + BN) * BK;           // elements per stage (A then B): 32 KiB
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 2, wn = wave & 3;
  const int ntile = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int nk = g.K >> 5;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;
  const int c4 = lane & 3, r16 = lane >> 2;
  const int frow = lane & 15, fk = lane >> 4;
  const int la = wm * 128 * BK + frow * BK + ((fk ^ swz64(frow)) << 3);
  const int lb = BM * BK + wn * 64 * BK + ((frow >> 2) * 16 + (frow & 3)) * BK + ((fk ^ swz64(frow)) << 3);

#pragma unroll 1
  for (int t0 = 0; t0 < ntile; t0 += G) {
    const int nchunk = min(G, ntile - t0);
    if ((int)blockIdx.x >= nchunk) break;
    const int tile = t0 + xcd_remap(blockIdx.x, nchunk);
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* ap[2];
    const bf16_t* bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 16 + r16;
      ap[j] = Ab + outer_off(g.A, min(m0 + row, g.M - 1)) + ((c4 ^ swz64(row)) << 3);
      bp[j] = Bb + outer_off(g.B, min(n0 + row, g.N - 1)) + ((c4 ^ swz64_b(row)) << 3);
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int off, int kt) {
      bf16_t* ad = smem + off + wave * 2 * 16 * BK;
      bf16_t* bd = smem + off + BM * BK + wave * 2 * 16 * BK;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * BK), (lvoid_t*)(ad + j * 16 * BK), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kt * BK), (lvoid_t*)(bd + j * 16 * BK), 16, 0, 0);
    };
    auto stage_piece = [&](int off, int kt, int j) {       // piece j of 4: A0, A1, B0, B1
      if (j < 2)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * BK),
                                         (lvoid_t*)(smem + off + (wave * 2 + j) * 16 * BK), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j - 2] + kt * BK),
                                         (lvoid_t*)(smem + off + BM * BK + (wave * 2 + j - 2) * 16 * BK), 16, 0, 0);
    };
    // wait until all but the `newer` youngest stages of this wave's DMA have landed (4 pieces per stage)
    auto wait_stages = [&](int newer) {
      if (newer >= 2) wait_vmcnt<8>(); else if (newer == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    };

    __builtin_amdgcn_s_barrier();          // every wave has finished reading the previous tile's stages
#pragma unroll
    for (int st = 0; st < S - 1; ++st)
      if (st < nk) stage(st * STAGE, st);
    wait_stages(min(nk, S - 1) - 1 > 2 ? 2 : min(nk, S - 1) - 1);
    __builtin_amdgcn_s_barrier();
    frag8_t af[FM], bcur[FN], bnext[FN], alast;
#pragma unroll
    for (int j = 0; j < FN; ++j) bcur[j] = *reinterpret_cast<const frag8_t*>(smem + lb + j * 4 * BK);
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const frag8_t*>(smem + la + i * 16 * BK);

    int nb = STAGE;                        // stage offset of step kt + 1
    int fb = (S - 1) * STAGE;              // stage offset that step kt + S - 1 is loaded into
#pragma unroll 1
    for (int kt = 0; kt < nk - 1; ++kt) {
      // stage kt+1 landed (own pieces), then everyone's: newer stages in flight = min(S - 3, nk - kt - 2)
      wait_stages(min(S - 3, nk - kt - 2));
      __builtin_amdgcn_s_barrier();
      // the four DMA pieces of stage kt+S-1 are spread over the MFMA groups: issued back to back behind the barrier
      // they hold BOTH waves of a SIMD in the (slow, queue-limited) DMA issue while the matrix pipe idles
      const bool do_stage = kt + S - 1 < nk;
      const bf16_t* pa = smem + nb + la;
      const bf16_t* pb = smem + nb + lb;
      // the next step's B fragments go out behind the first MFMA group: the compiler's wait in front of that group
      // then only covers reads issued a whole step ago
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<TE>(bcur[j], af[i], acc[i][j]);
        if (i == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; ++j) bnext[j] = *reinterpret_cast<const frag8_t*>(pb + j * 4 * BK);
          alast = *reinterpret_cast<const frag8_t*>(pa + (FM - 1) * 16 * BK);   // early: nothing may trail the last group
        }
        if (i < FM - 1) af[i] = *reinterpret_cast<const frag8_t*>(pa + i * 16 * BK);
        if ((i & 1) && do_stage) stage_piece(fb, kt + S - 1, i >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      af[FM - 1] = alast;
#pragma unroll
      for (int j = 0; j < FN; ++j) bcur[j] = bnext[j];
      nb = nb == (S - 1) * STAGE ? 0 : nb + STAGE;
      fb = fb == (S - 1) * STAGE ? 0 : fb + STAGE;
    }
    if (nk > 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<TE>(bcur[j], af[i], acc[i][j]);
    }

    TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
    TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
    const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
    const int nc = n0 + wn * 64 + fk * 16;
      float cv0[8], cv1[8];
    load_col8(g, bias, nc, cv0);
    load_col8(g, bias, nc + 8, cv1);
    if constexpr (sizeof(TC) == 2) {       // (the host sends 16-bit outputs here only where lines_ok holds)
      W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, FM>(g, Cz, auxz, acc, m0 + wm * 128, nc, lane, cv0, cv1, nullptr)));
    } else {
      W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, FM>(g, Cz, auxz, acc, m0 + wm * 128 + frow, nc, cv0, cv1)));
    }
  }   // tile loop
}

// ------------------------------------------------------------------------------ 256 x 256 x 64, phased (anti-phase wave groups)
// PMC on the two ring kernels above (tools/gemm_pmc.py): ~40 % of all wave cycles are parked at s_waitcnt / s_barrier
// and the matrix pipes are busy 25 % of the time.  Both waves of a SIMD run the same read -> wait -> MFMA sequence in
// lockstep behind the per-step barrier, so the pipe idles whenever they wait for LDS or for a DMA stage.
// This kernel schedules the two waves of every SIMD in ANTI-PHASE (cdna_hip_programming.md 5, "8-phase" structure):
//   * 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 (128 accumulator VGPRs); waves w and w + 4 share a SIMD and
//     form the two groups (m halves).  A K tile of 64 is worked off in FOUR phases of 16 MFMAs (A half x B half x 2
//     k-steps); a phase is  {ds_read the operands this phase needs | issue 2 DMA pieces | counted vmcnt} s_barrier
//     {lgkmcnt(0) | 16 MFMAs} s_barrier.  Group 1 runs ONE barrier behind group 0, so on every SIMD one wave multiplies
//     while the other reads: the matrix pipe always has a wave with its operands in registers.
//   * LDS = two K-tile buffers of 64 KiB ([256 A rows | 256 B rows] x 128 B, chunk-swizzled as in the ring kernels),
//     refilled by QUARTERS of 16 KiB in the order the phases consume them -- QA0 (first 64 rows of each group's A
//     half) and QB0 (B rows with row & 8 == 0) are read in phase 1, QB1 in phase 2, QA1 in phase 3 -- and each quarter
//     is re-issued two phases after its last read (strictly after BOTH groups' reads have returned): a quarter is in
//     flight for 5-6 phases (~1.5 us), four quarters at a time, `s_waitcnt vmcnt(8)`, never vmcnt(0) in steady state.
//   * same register epilogue as the 256x256 ring kernel (B rows permuted so a lane owns 16 consecutive columns).
template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// wait until all but the `newer` most recently issued quarters (2 DMA pieces each) of this wave have landed
__device__ __forceinline__ void wait_quarters(int newer) {
  if (newer >= 4) wait_vm<8>();
  else if (newer == 3) wait_vm<6>();
  else if (newer == 2) wait_vm<4>();
  else if (newer == 1) wait_vm<2>();
  else wait_vm<0>();
}

template <typename TE, typename TC>
__global__ __launch_bounds__(512) void gemm16_phased_256x256_kernel(const GemmArgs g) {
  constexpr int BM = 256, BN = 256, BK = 64, FM = 8, FN = 4;
  constexpr int BUF = (BM + BN) * BK;             // elements per K-tile buffer: A [256][64] then B [256][64]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wr = wave >> 2, wc = wave & 3;                     // wr = wave group (waves w, w + 4 share a SIMD)
  const int ntile = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int nk = g.K >> 6;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;
  const int c8 = lane & 7, r8 = lane >> 3;
  const int frow = lane & 15, fk = lane >> 4;
  // fragment offsets (elements) inside a buffer, k-step 0 / 1
  const int sw = swz(frow);
  const int la0 = (wr * 128 + frow) * 64 + ((fk ^ sw) << 3);
  const int la1 = (wr * 128 + frow) * 64 + (((4 + fk) ^ sw) << 3);
  const int brow = (frow >> 2) * 16 + (frow & 3);
  const int lb0 = BM * 64 + (wc * 64 + brow) * 64 + ((fk ^ sw) << 3);
  const int lb1 = BM * 64 + (wc * 64 + brow) * 64 + (((4 + fk) ^ sw) << 3);
  // DMA pieces of this wave: quarter q in {QA0, QB0, QB1, QA1}, piece j in {0, 1}; a piece = 8 consecutive LDS rows
  //   QA0: piece p < 8 -> A rows p*8 .., p >= 8 -> 128 + (p-8)*8 ..      QA1: the same + 64
  //   QB0: B rows p*16 ..                                                QB1: p*16 + 8 ..
  auto piece_row = [&](int q, int j) -> int {
    const int p = wave * 2 + j;
    if (q == 0) return (p < 8 ? p * 8 : 128 + (p - 8) * 8);
    if (q == 3) return (p < 8 ? p * 8 : 128 + (p - 8) * 8) + 64;
    return p * 16 + (q == 2 ? 8 : 0);
  };

#pragma unroll 1
  for (int t0 = 0; t0 < ntile; t0 += G) {
    const int nchunk = min(G, ntile - t0);
    if ((int)blockIdx.x >= nchunk) break;
    const int tile = t0 + xcd_remap(blockIdx.x, nchunk);
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // per-lane source pointers of the 8 pieces (K offset added at issue)
    // element offsets from the operand base (32 bits: the largest operand, conv1's input, has 3.2e8 elements) -- as
    // 64-bit pointers the eight sources cost 8 more VGPRs than this kernel has (it sits at the 256-register limit)
    int soff[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool isa = q == 0 || q == 3;
      int col[2];
      const bf16_t* ptr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = piece_row(q, j) + r8;
        col[j] = (c8 ^ (isa ? swz(row) : swz_b(row))) << 3;
      }
      // the two pieces of a quarter are 8 (A) / 16 (B) rows apart
      if (isa) tile_ptrs<2>(g.A, Ab, m0, BM, g.M, piece_row(q, 0) + r8, 8, col, ptr);
      else tile_ptrs<2>(g.B, Bb, n0, BN, g.N, piece_row(q, 0) + r8, 16, col, ptr);
#pragma unroll
      for (int j = 0; j < 2; ++j) soff[q][j] = (int)(ptr[j] - (isa ? Ab : Bb));
    }
    auto issue = [&](int q, int kt) {               // quarter q of K tile kt -> buffer kt & 1
      const bool isa = q == 0 || q == 3;
      bf16_t* base = smem + (kt & 1) * BUF + (isa ? 0 : BM * 64);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gvoid_t*)((isa ? Ab : Bb) + (soff[q][j] + kt * 64)),
                                         (lvoid_t*)(base + piece_row(q, j) * 64), 16, 0, 0);
    };
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K tile 0 entirely, QA0 / QB0 of K tile 1 (issue order = consumption order)
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    if (nk > 1) { issue(0, 1); issue(1, 1); }
    wait_quarters(2 + (nk > 1 ? 2 : 0));             // QA0(0), QB0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                    // ... everyone's
    if (wr == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind from here on

    frag8_t af[4][2], b0[2][2], b1[2][2];
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const bf16_t* bufp = smem + (kt & 1) * BUF;
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      // ---------------- phase 1: read B0 + A lo; issue QB1(kt+1); MFMA A lo x B0
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b0[j][0] = *reinterpret_cast<const frag8_t*>(bufp + lb0 + j * 4 * 64);
        b0[j][1] = *reinterpret_cast<const frag8_t*>(bufp + lb1 + j * 4 * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i][0] = *reinterpret_cast<const frag8_t*>(bufp + la0 + i * 16 * 64);
        af[i][1] = *reinterpret_cast<const frag8_t*>(bufp + la1 + i * 16 * 64);
      }
      if (more1) issue(2, kt + 1);
      wait_quarters(1 + (more1 ? 3 : 0));            // QB1(kt) for phase 2
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<TE>(b0[j][kk], af[i][kk], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 2: read B1; issue QA1(kt+1); MFMA A lo x B1
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b1[j][0] = *reinterpret_cast<const frag8_t*>(bufp + lb0 + (2 + j) * 4 * 64);
        b1[j][1] = *reinterpret_cast<const frag8_t*>(bufp + lb1 + (2 + j) * 4 * 64);
      }
      if (more1) issue(3, kt + 1);
      wait_quarters(more1 ? 4 : 0);                  // QA1(kt) for phase 3
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16<TE>(b1[j][kk], af[i][kk], acc[i][2 + j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 3: read A hi; issue QA0(kt+2); MFMA A hi x B1
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i][0] = *reinterpret_cast<const frag8_t*>(bufp + la0 + (4 + i) * 16 * 64);
        af[i][1] = *reinterpret_cast<const frag8_t*>(bufp + la1 + (4 + i) * 16 * 64);
      }
      if (more2) issue(0, kt + 2);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16<TE>(b1[j][kk], af[i][kk], acc[4 + i][2 + j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 4: (operands in registers); issue QB0(kt+2); MFMA A hi x B0
      if (more2) issue(1, kt + 2);
      if (more1) wait_quarters(2 + (more2 ? 2 : 0)); // QA0(kt+1), QB0(kt+1) for the next K tile's phase 1
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16<TE>(b0[j][kk], af[i][kk], acc[4 + i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();       // group 0 catches up: every read of this tile's buffers is done

    TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
    TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
    const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
    // the output row / column of this lane are made opaque HERE: left to itself the compiler forms the epilogue's
    // 64-bit row addresses before the main loop and, at the 256-register limit, spills them -- and a kernel that
    // touches scratch at all pays ~8 us per dispatch (tools/probes/scratch_probe.hip)
    int nc = n0 + wc * 64 + fk * 16, mr = m0 + wr * 128 + frow;
    asm volatile("" : "+v"(nc), "+v"(mr));
    float cv0[8], cv1[8];
    load_col8(g, bias, nc, cv0);
    load_col8(g, bias, nc + 8, cv1);
    if constexpr (sizeof(TC) == 2) {       // (the host sends 16-bit outputs here only where lines_ok holds)
      W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, FM>(g, Cz, auxz, acc, mr - frow, nc, lane, cv0, cv1, nullptr)));
    } else {
      W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, FM, 2>(g, Cz, auxz, acc, mr, nc, cv0, cv1)));
    }
  }   // tile loop
}

template <typename TE, typename TC>
static void launch_ph(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (256 + 256) * 64 * sizeof(bf16_t);   // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_phased_256x256_kernel<TE, TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 256);
  const int tiles = a.tiles_m * a.tiles_n;
  int ncu = 256;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  hipLaunchKernelGGL((gemm16_phased_256x256_kernel<TE, TC>), grid, dim3(512), lds, st, a);
}

// ------------------------------------------------------------------------------ 4-wave register-staged 256x256x64
// ONE wave per SIMD, 128 x 128 per wave (2 x 2 waves): 0.25 ds_read_b128 per v_mfma_f32_16x16x32 instead of the 0.375 /
// 0.5 of the 8-wave kernels above (their fragment reads + operand writes take the whole 128 B/clk LDS port for the 2048
// MFMA cycles of a K tile: 192 + 64 KB; here 128 + 64 KB).  With a single instruction stream per SIMD an LDS-DMA piece
// (~100 issue cycles each, 16 per K tile) would sit in front of the MFMAs (the round-2 4-wave kernel kept LDS-DMA and
// lost 20 %), so the operands are staged global_load_dwordx4 -> 64 VGPRs -> ds_write_b128.  The 256 accumulators are
// pinned in AGPRs by the asm form of the MFMA (mfma16_agpr); the main loop then uses 224 VGPRs with no spill.
// A K tile is two phases of 64 MFMAs in 32 fenced slots of two MFMAs + at most two other instructions:
//   phase 0: fragments of k-step 1 (16 ds_read) | 8 global loads A(K tile + 2) | staging of B(K tile + 1) -> other buffer
//   barrier  (the only one per K tile)
//   phase 1: fragments of the next K tile's k-step 0 | 8 global loads B(K tile + 2) | staging of A(K tile + 2)
// OPT-IN (W2V2_GEMM_QUAD=1 / family 5 of w2v2_tune_gemm_kernel): bit-equal to the phased kernel on every epilogue, and
// measured 0-10 % SLOWER than it (DESIGN.md section 4 has the per-shape table and the time attribution).
// Same LDS image, swizzles, B row order and epilogues as gemm16_phased_256x256_kernel.
template <typename TE, typename TC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm16_quad_256x256_kernel(const GemmArgs g) {
  constexpr int BM = 256, BN = 256, BK = 64, FM = 8;
  constexpr int BUF = (BM + BN) * BK;             // elements per K-tile buffer: A [256][64] then B [256][64]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // 0..3
  const int ntile = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int nk = g.K >> 6;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

#pragma unroll 1
  for (int t0 = 0; t0 < ntile; t0 += G) {
    const int nchunk = min(G, ntile - t0);
    if ((int)blockIdx.x >= nchunk) break;
    const int tile = t0 + xcd_remap(blockIdx.x, nchunk);
    // (divisors made opaque per tile: hoisted out of the tile loop, their reciprocals live in VGPRs across the main
    // loop, where there is none to spare)
    int tiles_n = g.tiles_n;
    asm volatile("" : "+s"(tiles_n));
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // the thread index is rebuilt per tile from the wave number (an SGPR) and the lane count, and made opaque:
    // everything derived from it below is recomputed per tile (a few VALU instructions) instead of being kept alive --
    // and spilled -- across the main loop and the pointer set-up; a kernel that touches scratch at all pays ~8 us per
    // dispatch (tools/probes/scratch_probe.hip)
    int tid;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid));
    tid += wave_s * 64;
    // staging: thread -> 16-byte chunk sc of rows srow + 32 u (u < 8) of each operand tile
    const int srow = tid >> 3, sc = tid & 7;
    int soa[8], sob[8];                              // 32-bit element offsets of the 16 source chunks (host: fits32)
    {
      // one row offset at a time (sched_barrier): the general form is a 64-bit division per row, and sixteen of them
      // interleaved need more registers than the wave has
      auto offs = [&](const OpDev& o, int t0r, int bound, int (&so)[8]) {
        bool fast = t0r + 256 <= bound;
        int64_t seg_base = 0;
        int first = t0r;
        int sl = (int)o.seg_len;                       // (row counts fit an int)
        asm volatile("" : "+s"(sl));
        if (sl > 0) {
          const int q0 = t0r / sl, q1 = (t0r + 255) / sl;
          fast = fast && q0 == q1;
          seg_base = (int64_t)q0 * o.seg_stride;
          first = t0r - q0 * sl;
        }
        if (fast) {
          const int b0 = (int)(seg_base + (int64_t)(first + srow) * o.ld) + (sc << 3);
          const int step = (int)(32 * o.ld);
#pragma unroll
          for (int u = 0; u < 8; ++u) so[u] = b0 + u * step;
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = min(t0r + srow + 32 * u, bound - 1);
            int64_t off = (int64_t)r * o.ld;
            if (sl > 0) {
              const int q = r / sl;
              off = (int64_t)q * o.seg_stride + (int64_t)(r - q * sl) * o.ld;
            }
            so[u] = (int)off + (sc << 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      offs(g.A, m0, g.M, soa);
      offs(g.B, n0, g.N, sob);
    }
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fk = lane >> 4;
    const int sw = swz(frow);
    // fragment offsets (elements) inside a buffer for k-step 0 / 1; A fragment i: + i * 16 * 64, B fragment (h, j):
    // + h * 64 * 64 + j * 4 * 64 (B rows are taken in the order (rho >> 2) * 16 + j * 4 + (rho & 3), see the phased kernel)
    const int la0 = (wr * 128 + frow) * 64 + ((fk ^ sw) << 3);
    const int la1 = (wr * 128 + frow) * 64 + (((4 + fk) ^ sw) << 3);
    const int brow = (frow >> 2) * 16 + (frow & 3);
    const int lb0 = BM * 64 + (wc * 128 + brow) * 64 + ((fk ^ sw) << 3);
    const int lb1 = BM * 64 + (wc * 128 + brow) * 64 + (((4 + fk) ^ sw) << 3);
    // LDS staging addresses: the swizzles depend on row bits 0..2 (A) / 0, 1, 4 (B) only, so one address per operand +
    // immediates for the rows srow + 32 u
    const int wa = (tid >> 3) * 64 + (((tid & 7) ^ swz(tid >> 3)) << 3);
    const int wb = BM * 64 + (tid >> 3) * 64 + (((tid & 7) ^ swz_b(tid >> 3)) << 3);
    f32x4 accL[FM][4], accR[FM][4];                  // columns wc * 128 + {0..63, 64..127}
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) accL[i][j] = accR[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Registers of the main loop: fragments a[8] + b[8] + bn[8] (a[i] is re-read for the next k-step as soon as row i
    // of the current one is done, the next B fragments need their own registers) = 96, staging ra[8] + rb[8] = 64.
    // Staging life times (tile T in buffer T & 1, the barrier sits at the end of phase 0):
    //   A(T): loaded at the start of phase 0 of K tile T - 2, written at the end of phase 1 of T - 2
    //   B(T): loaded at the start of phase 1 of K tile T - 2, written at the end of phase 0 of T - 1
    // -> both are ~1.8 phases in flight, both land in their buffer after the barrier that ends its last reads and before
    //    the barrier in front of the first read of tile T (phase 1 of T - 1).
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 ra[8], rb[8];
    frag8_t a[8], b[8], bn[8];
#define W2V2_Q_GLOAD(dst_, base_, so_, kt_)                                                               \
    _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                         \
      dst_[u] = *reinterpret_cast<const u32x4*>(base_ + (so_[u] + (kt_) * 64));
#define W2V2_Q_LWRITE1(src_, buf_, w_, u_)                                                                \
    *reinterpret_cast<u32x4*>(smem + (buf_) * BUF + w_ + (u_) * 32 * 64) = src_[u_];
#define W2V2_Q_LWRITE(src_, buf_, w_)                                                                     \
    W2V2_Q_LWRITE1(src_, buf_, w_, 0) W2V2_Q_LWRITE1(src_, buf_, w_, 1) W2V2_Q_LWRITE1(src_, buf_, w_, 2)  \
    W2V2_Q_LWRITE1(src_, buf_, w_, 3) W2V2_Q_LWRITE1(src_, buf_, w_, 4) W2V2_Q_LWRITE1(src_, buf_, w_, 5)  \
    W2V2_Q_LWRITE1(src_, buf_, w_, 6) W2V2_Q_LWRITE1(src_, buf_, w_, 7)
#define W2V2_Q_RA(buf_, la_, i_) (*reinterpret_cast<const frag8_t*>(smem + (buf_) * BUF + la_ + (i_) * 16 * 64))
#define W2V2_Q_RB(buf_, lb_, i_)                                                                          \
    (*reinterpret_cast<const frag8_t*>(smem + (buf_) * BUF + lb_ + ((i_) >> 2) * 64 * 64 + ((i_) & 3) * 4 * 64))
    // A phase is 32 slots of two MFMAs (row i_ of the wave tile x B fragments j_, 4 + j_) + at most two other
    // instructions, each slot fenced by a sched_barrier: with ONE wave per SIMD whatever is issued in a bunch (the
    // compiler's choice: all eight global loads with their address arithmetic in front of the first MFMA, four LDS
    // writes back to back) is issue time the matrix pipe idles through -- measured on conv1: +89 us for the loads and
    // +82 us for the writes on top of 479 us of MFMAs + prologue / epilogue (temporary variants with the loads / the
    // writes / the fragment reads / the MFMAs compiled out); one at a time they fit in the 12 free cycles of an MFMA.
#define W2V2_Q_SLOT(bc_, av_, i_, j_, OPS_)                                                               \
    {                                                                          \
      mfma16_agpr<TE>(bc_[j_], av_, accL[i_][j_]);                                                        \
      mfma16_agpr<TE>(bc_[4 + (j_)], av_, accR[i_][j_]);                                                  \
    }                                                                                                     \
    OPS_                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
#define W2V2_Q_GL1(dst_, base_, so_, kt_, u_)                                                             \
    dst_[u_] = *reinterpret_cast<const u32x4*>(base_ + (so_[u_] + (kt_) * 64));
#define W2V2_Q_FA(rbuf_, la_, i_) a[i_] = W2V2_Q_RA(rbuf_, la_, i_);
#define W2V2_Q_FA7(dst_, rbuf_, la_) dst_ = W2V2_Q_RA(rbuf_, la_, 7);
#define W2V2_Q_FB(bnx_, rbuf_, lb_, i_) bnx_[i_] = W2V2_Q_RB(rbuf_, lb_, i_);
    // one phase = 64 MFMAs of the current fragments (a[0..6], a7c_ x bc_) + the reads of the next ones: a[i] in place
    // once row i is done, the last row's fragment and the B fragments into their second copies (a7n_, bnx_) -- so the
    // last LDS instruction of a phase sits six MFMAs before its end and the lgkmcnt(0) in front of the barrier / of the
    // next phase's first MFMA finds nothing to wait for; the 8 global loads of gdst_ in rows 0..3, the 8 LDS writes of
    // wsrc_ in rows 4..6
#define W2V2_Q_PHASE(bc_, bnx_, a7c_, a7n_, rbuf_, la_, lb_, gdst_, gbase_, gso_, gkt_, wsrc_, wbuf_, w_) \
    W2V2_Q_SLOT(bc_, a[0], 0, 0, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 0))                                \
    W2V2_Q_SLOT(bc_, a[0], 0, 1, W2V2_Q_FB(bnx_, rbuf_, lb_, 0))                                          \
    W2V2_Q_SLOT(bc_, a[0], 0, 2, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 1))                                \
    W2V2_Q_SLOT(bc_, a[0], 0, 3, W2V2_Q_FB(bnx_, rbuf_, lb_, 1))                                          \
    W2V2_Q_SLOT(bc_, a[1], 1, 0, W2V2_Q_FA(rbuf_, la_, 0))                                                \
    W2V2_Q_SLOT(bc_, a[1], 1, 1, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 2))                                \
    W2V2_Q_SLOT(bc_, a[1], 1, 2, W2V2_Q_FB(bnx_, rbuf_, lb_, 2))                                          \
    W2V2_Q_SLOT(bc_, a[1], 1, 3, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 3) W2V2_Q_FB(bnx_, rbuf_, lb_, 3)) \
    W2V2_Q_SLOT(bc_, a[2], 2, 0, W2V2_Q_FA(rbuf_, la_, 1))                                                \
    W2V2_Q_SLOT(bc_, a[2], 2, 1, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 4))                                \
    W2V2_Q_SLOT(bc_, a[2], 2, 2, W2V2_Q_FB(bnx_, rbuf_, lb_, 4))                                          \
    W2V2_Q_SLOT(bc_, a[2], 2, 3, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 5) W2V2_Q_FB(bnx_, rbuf_, lb_, 5)) \
    W2V2_Q_SLOT(bc_, a[3], 3, 0, W2V2_Q_FA(rbuf_, la_, 2))                                                \
    W2V2_Q_SLOT(bc_, a[3], 3, 1, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 6))                                \
    W2V2_Q_SLOT(bc_, a[3], 3, 2, W2V2_Q_FB(bnx_, rbuf_, lb_, 6))                                          \
    W2V2_Q_SLOT(bc_, a[3], 3, 3, W2V2_Q_GL1(gdst_, gbase_, gso_, gkt_, 7) W2V2_Q_FB(bnx_, rbuf_, lb_, 7)) \
    W2V2_Q_SLOT(bc_, a[4], 4, 0, W2V2_Q_FA(rbuf_, la_, 3))                                                \
    W2V2_Q_SLOT(bc_, a[4], 4, 1, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 0))                                     \
    W2V2_Q_SLOT(bc_, a[4], 4, 2, W2V2_Q_FA7(a7n_, rbuf_, la_))                                            \
    W2V2_Q_SLOT(bc_, a[4], 4, 3, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 1))                                     \
    W2V2_Q_SLOT(bc_, a[5], 5, 0, W2V2_Q_FA(rbuf_, la_, 4))                                                \
    W2V2_Q_SLOT(bc_, a[5], 5, 1, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 2))                                     \
    W2V2_Q_SLOT(bc_, a[5], 5, 2, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 3))                                     \
    W2V2_Q_SLOT(bc_, a[5], 5, 3, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 4))                                     \
    W2V2_Q_SLOT(bc_, a[6], 6, 0, W2V2_Q_FA(rbuf_, la_, 5))                                                \
    W2V2_Q_SLOT(bc_, a[6], 6, 1, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 5))                                     \
    W2V2_Q_SLOT(bc_, a[6], 6, 2, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 6))                                     \
    W2V2_Q_SLOT(bc_, a[6], 6, 3, W2V2_Q_LWRITE1(wsrc_, wbuf_, w_, 7))                                     \
    W2V2_Q_SLOT(bc_, a7c_, 7, 0, W2V2_Q_FA(rbuf_, la_, 6))                                                \
    W2V2_Q_SLOT(bc_, a7c_, 7, 1, )                                                                        \
    W2V2_Q_SLOT(bc_, a7c_, 7, 2, )                                                                        \
    W2V2_Q_SLOT(bc_, a7c_, 7, 3, )
    // ---- prologue: K tile 0 in buffer 0, A(1) in buffer 1, B(1) in rb, fragments of (tile 0, k-step 0)
    const int k1 = min(1, nk - 1);
    __builtin_amdgcn_s_barrier();                    // every wave has left the previous tile's buffers
    {
      u32x4 ta[8];                                   // A(1) travels with K tile 0 (one global latency, not two)
      W2V2_Q_GLOAD(ra, Ab, soa, 0);
      W2V2_Q_GLOAD(rb, Bb, sob, 0);
      W2V2_Q_GLOAD(ta, Ab, soa, k1);
      W2V2_Q_LWRITE(ra, 0, wa)
      W2V2_Q_LWRITE(rb, 0, wb)
      W2V2_Q_GLOAD(rb, Bb, sob, k1);                 // (first needed in row 4 of the first phase)
      W2V2_Q_LWRITE(ta, 1, wa)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = W2V2_Q_RA(0, la0, i); b[i] = W2V2_Q_RB(0, lb0, i); }
    frag8_t a7x = a[7], a7y = a[7];

#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1, oth = cur ^ 1;
      const int kg = min(kt + 2, nk - 1);            // (the last two K tiles re-load the last one: never consumed)
      // ---------------- phase 0: k-step 0 of tile kt; reads k-step 1; A(kt+2) -> ra; rb = B(kt+1) -> other buffer
      W2V2_Q_PHASE(b, bn, a7x, a7y, cur, la1, lb1, ra, Ab, soa, kg, rb, oth, wb)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // ---------------- phase 1: k-step 1 of tile kt; reads k-step 0 of tile kt+1; B(kt+2) -> rb; ra = A(kt+2) -> cur
      W2V2_Q_PHASE(bn, b, a7y, a7x, oth, la0, lb0, rb, Bb, sob, kg, ra, cur, wa)
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have written their AGPRs (asm: no hazard tracking)
#undef W2V2_Q_SLOT
#undef W2V2_Q_GL1
#undef W2V2_Q_FA
#undef W2V2_Q_FA7
#undef W2V2_Q_FB
#undef W2V2_Q_LWRITE1
#undef W2V2_Q_PHASE
#undef W2V2_Q_RA
#undef W2V2_Q_RB
#undef W2V2_Q_GLOAD
#undef W2V2_Q_LWRITE

    TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
    TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
    const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
    int nc = n0 + wc * 128 + fk * 16, mr = m0 + wr * 128 + frow;
    asm volatile("" : "+v"(nc), "+v"(mr));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float cv0[8], cv1[8];
      load_col8(g, bias, nc + 64 * h, cv0);
      load_col8(g, bias, nc + 64 * h + 8, cv1);
      if constexpr (sizeof(TC) == 2) {       // (the host sends 16-bit outputs here only where lines_ok holds)
        W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, FM>(g, Cz, auxz, h ? accR : accL, mr - frow, nc + 64 * h, lane, cv0, cv1,
                                                       nullptr)));
      } else {
        W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, FM, 2>(g, Cz, auxz, h ? accR : accL, mr, nc + 64 * h, cv0, cv1)));
      }
    }
  }   // tile loop
}

template <typename TE, typename TC>
static void launch_quad(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (256 + 256) * 64 * sizeof(bf16_t);   // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_quad_256x256_kernel<TE, TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 256);
  const int tiles = a.tiles_m * a.tiles_n;
  int ncu = 256;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  hipLaunchKernelGGL((gemm16_quad_256x256_kernel<TE, TC>), grid, dim3(256), lds, st, a);
}

static const bool g_w2v2_no_glds = getenv("W2V2_NO_GLDS") != nullptr;   // A/B switches for benchmarking
static const bool g_w2v2_glds3 = getenv("W2V2_NO_GLDS3") == nullptr;
static const bool g_w2v2_ph = getenv("W2V2_NO_GEMM_PH") == nullptr;
static const bool g_w2v2_tile256 = getenv("W2V2_NO_GLDS4") == nullptr;          // 256x256 tiles at all
static const bool g_w2v2_persistent = getenv("W2V2_G3_NONPERSISTENT") == nullptr;
static const int g_w2v2_g3n = getenv("W2V2_G3N") ? atoi(getenv("W2V2_G3N")) : 512;  // smallest N of the 256x128 kernel

static int g_w2v2_ncu = 0;
static int device_cus() {
  if (g_w2v2_ncu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&g_w2v2_ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_w2v2_ncu <= 0)
      g_w2v2_ncu = 256;
    // data-parallel runs: the persistent ring kernels fill every CU's registers, so RCCL's all-reduce workgroups
    // (side stream) only run between them; W2V2_RESERVE_CUS=n keeps n CUs out of the persistent grids
    const char* r = getenv("W2V2_RESERVE_CUS");
    if (r && atoi(r) > 0 && atoi(r) < g_w2v2_ncu) g_w2v2_ncu -= atoi(r);
  }
  return g_w2v2_ncu;
}

template <typename TE, typename TC>
static void launch_glds4(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)4 * (256 + 256) * 32 * sizeof(bf16_t);   // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_ring_256x256_kernel<TE, TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 256);
  const int tiles = a.tiles_m * a.tiles_n, ncu = device_cus();
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  hipLaunchKernelGGL((gemm16_ring_256x256_kernel<TE, TC>), grid, dim3(512), lds, st, a);
}

template <typename TE, typename TC>
static void launch_glds3(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * (256 + 128) * 64 * sizeof(bf16_t);   // 144 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_ring_256x128_kernel<TE, TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 128);
  const int ncu = g_w2v2_persistent ? device_cus() : (1 << 30);
  const int tiles = a.tiles_m * a.tiles_n;
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  hipLaunchKernelGGL((gemm16_ring_256x128_kernel<TE, TC>), grid, dim3(512), lds, st, a);
}

template <typename TE, int FM, int FN, typename TC>
static void launch_glds(const GemmArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (32 * FM + 32 * FN) * 64 * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_dma_128_kernel<TE, FM, FN, TC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm16_dma_128_kernel<TE, FM, FN, TC>), grid, dim3(256), lds, st, a);
}

// ------------------------------------------------------------------------------ exact f32 kernel
template <typename TC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, 4x4 outputs each
  const int tile = blockIdx.x;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * 64, n0 = tn * 64;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int split = blockIdx.y;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const float* Ab = reinterpret_cast<const float*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const float* Bb = reinterpret_cast<const float*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  float acc[4][4] = {};
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    // each thread stages 4 elements of A and of B; index so that the contiguous (inner) dimension
    // of the operand runs over consecutive threads
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;  // 0..1023
      {
        int r, k;
        if (g.A.trans) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
        float v = 0.f;
        if (m0 + r < g.M && k0 + k < kend)
          v = g.A.trans ? Ab[outer_off(g.A, k0 + k) + m0 + r] : Ab[outer_off(g.A, m0 + r) + k0 + k];
        As[k][r] = v;
      }
      {
        int r, k;
        if (g.B.trans) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
        float v = 0.f;
        if (n0 + r < g.N && k0 + k < kend)
          v = g.B.trans ? Bb[outer_off(g.B, k0 + k) + n0 + r] : Bb[outer_off(g.B, n0 + r) + k0 + k];
        Bs[k][r] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const TC* auxz = g.aux ? reinterpret_cast<const TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  TC* auxo = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    epilogue_store4<TC>(g, Cz, auxz, auxo, bias, m0 + ty * 4 + i, n0 + tx * 4, acc[i], split == 0);
}

// ------------------------------------------------------------------------------ exact f32 on the matrix cores
// v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate, bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md 3,
// "FP32-input MFMA") at the f32 vector RATE -- but issued by one instruction per 4096 multiply-adds instead of 64, with
// one VGPR per operand, so an untuned LDS-tiled kernel already runs ~2.4x a VALU tile kernel (the 64x64x16 kernel above,
// kept behind W2V2_F32_VALU for A/B).  This is the GEMM of the exact-f32 parity mode and of BASELINE configs[4]
// (ECAPA-TDNN at the reference's `precision: 32`, "MFMA off" = no reduced-precision matrix path: the numerics ARE f32).
//   128 x 128 x 16 block tile, 4 waves as 2 x 2, 64 x 64 per wave = 2 x 2 MFMA blocks (64 accumulator VGPRs);
//   both operands are staged K-MAJOR in LDS ([k][row], pitch 132: a fragment is 32 consecutive rows of one k ->
//   conflict-free ds_read_b32), the next K tile's global loads are in flight under the 32 MFMAs of the current one;
//   operands swapped (D[n][m]) so a lane holds 4 consecutive n per accumulator quad -> the shared 4-wide epilogue.
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(const GemmArgs g) {
  // BK = 32: a K-contiguous operand row contributes one whole 128-byte line per K tile (BK = 16 fetched half lines, and
  // the 2 x 16 KB of half-used lines per tile thrashed the 32 KB L1: 26 TFLOP/s).  LDS pitch: 132 words for K-major
  // sources (16-byte aligned float4 stores), 129 for K-contiguous ones (their transposing scalar stores hit
  // (k + row) % 32 -> 2-way instead of 4-way conflicts); fragment reads [k][32 consecutive rows] are conflict-free
  // with either.
  constexpr int BM = 128, BN = 128, BK = 32, PA = TA ? 132 : 129, PB = TB ? 132 : 129;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];            // 2 x 32 x (PA + PB) floats = 66-68 KB
  float (*As)[BK][PA] = reinterpret_cast<float (*)[BK][PA]>(smem_raw);
  float (*Bs)[BK][PB] = reinterpret_cast<float (*)[BK][PB]>(smem_raw + sizeof(float) * 2 * BK * PA);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int split = blockIdx.y;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const float* Ab = reinterpret_cast<const float*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const float* Bb = reinterpret_cast<const float*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  // staging map: 4 x float4 per operand and thread.  K-contiguous operand (trans = 0): thread -> (row = c >> 3, 4 k):
  // eight lanes read one 128-byte line; K-major operand (trans = 1): thread -> (k = c >> 5, 4 rows)
  auto load_op = [&](const OpDev& o, const float* __restrict__ base, auto trans_c, int r0, int rbound, int k0,
                     float4 (&reg)[4]) {
    constexpr bool trans = decltype(trans_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!trans) {
        const int row = c >> 3, k = k0 + (c & 7) * 4;
        if (r0 + row < rbound && k < kend) {
          const float* p = base + outer_off(o, r0 + row) + k;
          if (o.vec_ok && k + 4 <= kend) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < kend) v[e] = p[e];
          }
        }
      } else {
        const int k = k0 + (c >> 5), row = (c & 31) * 4;
        if (k < kend && r0 + row < rbound) {
          const float* p = base + outer_off(o, k) + r0 + row;
          if (o.vec_ok && r0 + row + 4 <= rbound) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (r0 + row + e < rbound) v[e] = p[e];
          }
        }
      }
      reg[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_a = [&](int buf, const float4 (&reg)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      if constexpr (!TA) {
        const int row = c >> 3, k = (c & 7) * 4;
        As[buf][k][row] = reg[j].x; As[buf][k + 1][row] = reg[j].y; As[buf][k + 2][row] = reg[j].z; As[buf][k + 3][row] = reg[j].w;
      } else {
        *reinterpret_cast<float4*>(&As[buf][c >> 5][(c & 31) * 4]) = reg[j];
      }
    }
  };
  auto store_b = [&](int buf, const float4 (&reg)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      if constexpr (!TB) {
        const int row = c >> 3, k = (c & 7) * 4;
        Bs[buf][k][row] = reg[j].x; Bs[buf][k + 1][row] = reg[j].y; Bs[buf][k + 2][row] = reg[j].z; Bs[buf][k + 3][row] = reg[j].w;
      } else {
        *reinterpret_cast<float4*>(&Bs[buf][c >> 5][(c & 31) * 4]) = reg[j];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  float4 ra[4], rb[4];
  const std::integral_constant<bool, TA> ta_c{};
  const std::integral_constant<bool, TB> tb_c{};
  if (nk > 0) {
    load_op(g.A, Ab, ta_c, m0, g.M, kbeg, ra);
    load_op(g.B, Bb, tb_c, n0, g.N, kbeg, rb);
    store_a(0, ra);
    store_b(0, rb);
  }
  __syncthreads();
  const int kl = lane >> 5, rl = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_op(g.A, Ab, ta_c, m0, g.M, kbeg + (kt + 1) * BK, ra);
      load_op(g.B, Bb, tb_c, n0, g.N, kbeg + (kt + 1) * BK, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[cur][kk + kl][wm * 64 + i * 32 + rl];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kk + kl][wn * 64 + j * 32 + rl];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);   // D[n][m]
    }
    if (kt + 1 < nk) {
      store_a(cur ^ 1, ra);
      store_b(cur ^ 1, rb);
    }
    __syncthreads();
  }

  float* Cz = reinterpret_cast<float*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const float* auxz = g.aux ? reinterpret_cast<const float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  float* auxo = g.aux ? reinterpret_cast<float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  // D[n][m]: lane l, register r: m = l & 31, n = 8 (r >> 2) + 4 (l >> 5) + (r & 3)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        epilogue_store4<float>(g, Cz, auxz, auxo, bias, m0 + wm * 64 + i * 32 + rl, n0 + wn * 64 + j * 32 + q * 8 + kl * 4,
                               v4, split == 0);
      }
}

// ------------------------------------------------------------------------------ host dispatch
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename TE, int FM, int FN, typename TC>
static void launch_bf16(const GemmArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (32 * FM + 32 * FN) * 64 * sizeof(bf16_t);
  const bool ta = a.A.trans, tb = a.B.trans;
#define W2V2_LAUNCH(TA_, TB_)                                                                   \
  do {                                                                                          \
    static bool attr_set = false;                                                               \
    if (!attr_set) {                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_regstage_kernel<TE, FM, FN, TA_, TB_, TC>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
      attr_set = true;                                                                          \
    }                                                                                           \
    hipLaunchKernelGGL((gemm16_regstage_kernel<TE, FM, FN, TA_, TB_, TC>), grid, dim3(256), lds, st, a);  \
  } while (0)
  if (!ta && !tb) W2V2_LAUNCH(false, false);
  else if (!ta && tb) W2V2_LAUNCH(false, true);
  else if (ta && !tb) W2V2_LAUNCH(true, false);
  else W2V2_LAUNCH(true, true);
#undef W2V2_LAUNCH
}

// tuning hook (tools/gemm_shapes.py): force the kernel family of the K-contiguous 16-bit products.
//   0 = dispatch below, 1 = 128x128 LDS-DMA, 2 = 256x128 ring, 3 = 256x256x32 ring, 4 = 256x256x64 phased,
//   5 = 256x256x64 4-wave register-staged
static int g_w2v2_force = 0;
static const bool g_w2v2_quad = getenv("W2V2_GEMM_QUAD") != nullptr;    // A/B: 4-wave kernel wherever the phased one runs
extern "C" int w2v2_tune_gemm_kernel(int family) {
  const int old = g_w2v2_force;
  if (family >= 0 && family <= 5) g_w2v2_force = family;
  return old;
}

extern "C" int w2v2_gemm(const w2v2_gemm_desc* d, void* stream) {
  W2V2_REQUIRE(d != nullptr, "w2v2_gemm: null descriptor");
  W2V2_REQUIRE(d->M > 0 && d->N > 0 && d->K >= 0 && d->batch > 0, "w2v2_gemm: bad shape M=%d N=%d K=%d batch=%d",
               d->M, d->N, d->K, d->batch);
  W2V2_REQUIRE(d->A.ptr && d->B.ptr && d->C, "w2v2_gemm: null operand");
  W2V2_REQUIRE(d->dtype_ab == W2V2_F32 || d->dtype_ab == W2V2_BF16 || d->dtype_ab == W2V2_F16,
               "w2v2_gemm: bad dtype_ab %d", d->dtype_ab);
  W2V2_REQUIRE(d->dtype_c == W2V2_F32 || d->dtype_c == d->dtype_ab, "w2v2_gemm: dtype_c %d must be f32 or dtype_ab (%d)",
               d->dtype_c, d->dtype_ab);
  W2V2_REQUIRE(d->epilogue >= 0 && d->epilogue <= W2V2_EPI_MUL, "w2v2_gemm: bad epilogue %d", d->epilogue);
  const int split = d->split_k > 1 ? d->split_k : 1;
  const int atomic = (split > 1 || d->accumulate) ? 1 : 0;
  if (atomic) {
    W2V2_REQUIRE(d->dtype_c == W2V2_F32, "w2v2_gemm: split_k/accumulate need an f32 C");
    W2V2_REQUIRE(d->epilogue == W2V2_EPI_NONE || (split == 1) || d->epilogue == W2V2_EPI_BIAS,
                 "w2v2_gemm: split_k supports EPI_NONE/EPI_BIAS only");
  }
  if (d->epilogue == W2V2_EPI_BIAS || d->epilogue == W2V2_EPI_BIAS_GELU || d->epilogue == W2V2_EPI_BIAS_GELU_GRAD)
    W2V2_REQUIRE(d->bias != nullptr, "w2v2_gemm: bias epilogue without bias");
  if (d->epilogue == W2V2_EPI_GELU_BWD || d->epilogue == W2V2_EPI_ADD || d->epilogue == W2V2_EPI_MUL ||
      d->epilogue == W2V2_EPI_BIAS_GELU_GRAD)
    W2V2_REQUIRE(d->aux != nullptr, "w2v2_gemm: epilogue %d needs aux", d->epilogue);
  if (d->epilogue == W2V2_EPI_SCALE_RC)
    W2V2_REQUIRE(d->row_scale && d->col_scale, "w2v2_gemm: EPI_SCALE_RC needs row/col scales");

  GemmArgs a;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.epilogue = d->epilogue; a.split_k = split; a.atomic = atomic;
  a.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
  auto cvt = [&](const w2v2_operand& o, int64_t s0, int64_t s1) {
    OpDev r;
    r.ptr = o.ptr; r.ld = o.ld; r.seg_len = o.seg_len; r.seg_stride = o.seg_stride; r.trans = o.trans ? 1 : 0;
    r.vec_ok = aligned16(o.ptr) && (o.ld % 8 == 0) && (o.seg_stride % 8 == 0) && (s0 % 8 == 0) && (s1 % 8 == 0);
    return r;
  };
  a.A = cvt(d->A, d->A.stride0, d->A.stride1);
  a.B = cvt(d->B, d->B.stride0, d->B.stride1);
  a.a_s0 = d->A.stride0; a.a_s1 = d->A.stride1; a.b_s0 = d->B.stride0; a.b_s1 = d->B.stride1;
  a.C = d->C; a.ldc = d->ldc; a.c_s0 = d->c_stride0; a.c_s1 = d->c_stride1;
  a.aux = d->aux; a.ldaux = d->ldaux; a.aux_s0 = d->aux_stride0; a.aux_s1 = d->aux_stride1;
  a.bias = d->bias; a.bias_s1 = d->bias_stride1;
  a.row_scale = d->row_scale; a.col_scale = d->col_scale;
  a.alpha = d->alpha;
  a.k_ext = d->k_ext; a.n_ext_from = d->n_ext_from; a.b_lo_off = d->b_lo_offset;
  static const bool defer_env = getenv("W2V2_NO_DEFER") == nullptr;   // A/B switch
  a.defer_ok = 0;
  const int cal = d->dtype_c == W2V2_F32 ? 4 : 8;     // elements per 16 bytes
  a.c_vec_ok = aligned16(d->C) && (d->ldc % cal == 0) && (d->c_stride0 % cal == 0) && (d->c_stride1 % cal == 0);
  a.aux_vec_ok = d->aux && aligned16(d->aux) && (d->ldaux % cal == 0) && (d->aux_stride0 % cal == 0) &&
                 (d->aux_stride1 % cal == 0);
  hipStream_t st = as_stream(stream);

  if (d->dtype_ab != W2V2_F32) {
    const bool narrow = d->N <= 64;
    const int BM = 128, BN = narrow ? 64 : 128;
    a.tiles_m = (int)cdiv(d->M, BM); a.tiles_n = (int)cdiv(d->N, BN);
    a.k_per_split = (int)(cdiv(cdiv(d->K, split), 64) * 64);
    if (a.k_per_split == 0) a.k_per_split = 64;
    dim3 grid(a.tiles_m * a.tiles_n, split, d->batch);
    const bool glds = !a.A.trans && !a.B.trans && a.A.vec_ok && a.B.vec_ok && (d->K % 64 == 0) && d->K >= 64 &&
                      !g_w2v2_no_glds;
    // 256x128 3-stage kernel for the encoder shapes; the conv stack (N = 512, M ~ 3e5) measures
    // slightly faster on the 128x128 kernel at 2 workgroups per CU
    // (two-term weights exist on this kernel only: such a request takes it whatever the shape)
    const bool big = glds && split == 1 && !atomic &&
                     (d->k_ext != 0 || (d->N >= g_w2v2_g3n && d->M >= 1024 && g_w2v2_glds3));
    // 256x256 tiles when they fill the chip: >= 85 % of the CU slots of the last round busy (FFN1, dH, conv stack)
    bool huge = false;
    if (big && d->k_ext == 0 && d->N >= 512 && d->batch == 1 && g_w2v2_tile256) {
      const int64_t t4 = cdiv(d->M, 256) * cdiv(d->N, 256), ncu = device_cus();
      // the phased kernel keeps its DMA sources as 32-bit element offsets from the operand base
      auto extent = [](const w2v2_operand& o, int64_t rows, int64_t K) -> int64_t {
        return (o.seg_len > 0 ? (rows / o.seg_len + 1) * o.seg_stride + o.seg_len * o.ld : rows * o.ld) + K;
      };
      const bool fits32 = extent(d->A, d->M, d->K) < (int64_t(1) << 31) && extent(d->B, d->N, d->K) < (int64_t(1) << 31);
      // measured per shape with tools/gemm_shapes.py (FAMILIES=0,1,2,3,4): a 256x256 tile step runs ~1.2x the flops
      // per second of a 256x128 one (1.25 below), so the larger tile wins whenever its last-round fill is not worse by more than
      // that (conv3: 620 tiles = 0.81 vs 0.97 -> 174 vs 183 us; conv5: 156 tiles = 0.61 vs 0.61 -> 41 vs 47 us;
      // FFN2-shaped N = 768 products: 117 tiles = 0.46 vs 0.91 -> 73 vs 52 us stay on the 256x128 ring)
      const int64_t t3 = cdiv(d->M, 256) * cdiv(d->N, 128);
      const double fill4 = (double)t4 / (double)(cdiv(t4, ncu) * ncu), fill3 = (double)t3 / (double)(cdiv(t3, ncu) * ncu);
      // 16-bit outputs of the 256x256 kernels go through the full-line register epilogue only
      const bool lines = d->dtype_c == W2V2_F32 ||
                         (a.c_vec_ok && (d->N % 64 == 0) && (d->aux == nullptr || a.aux_vec_ok));
      huge = lines && fill4 * 1.25 >= fill3 && t4 * 2 >= ncu &&
             (double)d->N / (double)(cdiv(d->N, 256) * 256) >= 0.9 && (fits32 || !g_w2v2_ph);
    }
    bool big_ = big, force_ph = true, quad = g_w2v2_quad;
    if (g_w2v2_force != 0 && glds && split == 1 && !atomic && d->k_ext == 0 && d->batch == 1) {
      big_ = g_w2v2_force >= 2;
      huge = g_w2v2_force >= 3 && (d->dtype_c == W2V2_F32 ||
                                   (a.c_vec_ok && (d->N % 64 == 0) && (d->aux == nullptr || a.aux_vec_ok)));
      force_ph = g_w2v2_force >= 4;
      quad = g_w2v2_force == 5;
    }
    if (d->k_ext != 0)
      W2V2_REQUIRE(big && !huge && d->k_ext == d->K && d->n_ext_from >= 0 && d->n_ext_from % 128 == 0 &&
                       d->b_lo_offset % 8 == 0,
                   "w2v2_gemm: two-term weights (k_ext) need k_ext == K, n_ext_from %% 128 == 0 and K-contiguous 16-byte "
                   "aligned operands with K %% 64 == 0 (M=%d N=%d K=%d)", d->M, d->N, d->K);
    // TE = operand element type (selects the MFMA instruction), TC = float or TE
#define W2V2_GEMM_LAUNCH(TE, TC)                                                                         \
    do {                                                                                                 \
      if (huge && g_w2v2_ph && force_ph && quad) launch_quad<TE, TC>(a, d->M, d->N, d->batch, st);        \
      else if (huge && g_w2v2_ph && force_ph) launch_ph<TE, TC>(a, d->M, d->N, d->batch, st);             \
      else if (huge) launch_glds4<TE, TC>(a, d->M, d->N, d->batch, st);                                  \
      else if (big_) launch_glds3<TE, TC>(a, d->M, d->N, d->batch, st);                                  \
      else if (glds) { if (narrow) launch_glds<TE, 4, 2, TC>(a, grid, st); else launch_glds<TE, 4, 4, TC>(a, grid, st); } \
      else { if (narrow) launch_bf16<TE, 4, 2, TC>(a, grid, st); else launch_bf16<TE, 4, 4, TC>(a, grid, st); }         \
    } while (0)
    if (big_ && !huge)
      a.defer_ok = defer_env && d->dtype_c != W2V2_F32 && a.c_vec_ok && (d->N % 128 == 0) && !atomic &&
                   (d->aux == nullptr || a.aux_vec_ok) &&
                   d->epilogue != W2V2_EPI_BIAS_GELU && d->epilogue != W2V2_EPI_BIAS_GELU_GRAD;
    if (d->dtype_ab == W2V2_BF16) {
      if (d->dtype_c == W2V2_F32) W2V2_GEMM_LAUNCH(bf16_t, float); else W2V2_GEMM_LAUNCH(bf16_t, bf16_t);
    } else {
      if (d->dtype_c == W2V2_F32) W2V2_GEMM_LAUNCH(f16_t, float); else W2V2_GEMM_LAUNCH(f16_t, f16_t);
    }
#undef W2V2_GEMM_LAUNCH
  } else {
    W2V2_REQUIRE(d->dtype_c == W2V2_F32, "w2v2_gemm: f32 operands need an f32 C");
    static const bool f32_valu = getenv("W2V2_F32_VALU") != nullptr;      // A/B: the 64x64x16 VALU tile kernel
    const int BT = f32_valu ? 64 : 128;
    a.tiles_m = (int)cdiv(d->M, BT); a.tiles_n = (int)cdiv(d->N, BT);
    a.k_per_split = (int)(cdiv(cdiv(d->K, split), 32) * 32);
    if (a.k_per_split == 0) a.k_per_split = 32;
    // f32 rows are 16-byte vectors of FOUR elements
    auto vec4 = [&](const w2v2_operand& o) {
      return aligned16(o.ptr) && (o.ld % 4 == 0) && (o.seg_stride % 4 == 0) && (o.stride0 % 4 == 0) && (o.stride1 % 4 == 0);
    };
    a.A.vec_ok = vec4(d->A); a.B.vec_ok = vec4(d->B);
    dim3 grid(a.tiles_m * a.tiles_n, split, d->batch);
#define W2V2_F32_LAUNCH(TA_, TB_)                                                                                     \
    do {                                                                                                              \
      constexpr size_t lds = sizeof(float) * 2 * 32 * ((TA_ ? 132 : 129) + (TB_ ? 132 : 129));                          \
      static bool attr_set = false;                                                                                   \
      if (!attr_set) {                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<TA_, TB_>),                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        attr_set = true;                                                                                              \
      }                                                                                                               \
      hipLaunchKernelGGL((gemm_f32_mfma_kernel<TA_, TB_>), grid, dim3(256), lds, st, a);                               \
    } while (0)
    if (f32_valu) hipLaunchKernelGGL(gemm_f32_kernel<float>, grid, dim3(256), 0, st, a);
    else if (!a.A.trans && !a.B.trans) W2V2_F32_LAUNCH(false, false);
    else if (!a.A.trans && a.B.trans) W2V2_F32_LAUNCH(false, true);
    else if (a.A.trans && !a.B.trans) W2V2_F32_LAUNCH(true, false);
    else W2V2_F32_LAUNCH(true, true);
#undef W2V2_F32_LAUNCH
  }
  W2V2_CHECK_LAUNCH("w2v2_gemm");
  return 0;
}
