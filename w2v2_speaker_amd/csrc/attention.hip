// attention.hip -- fused multi-head self-attention for ANY sequence length (head dim 64): HF:438-548 minus the
// q/k/v/out projections.  16-bit in (bf16 or fp16), f32 softmax, 16-bit out.
//
// One skeleton for the three kernels.  A workgroup owns 64 "rows" (queries in the forward / dQ kernel, keys in the
// dK/dV kernel), 16 per wave, whose operand fragments live in registers; the OTHER sequence axis streams through LDS in
// tiles of 64 rows (double-buffered row-major images, 16-B chunks XOR-swizzled, next tile's global loads in flight under
// the current tile's MFMAs).  Scores are computed with swapped MFMA operands (D[row = n][col = m]) so each lane holds,
// for ONE of its 16 rows, 4 consecutive columns per 16x16 fragment: the softmax row reductions are in-register + two wave
// shuffles (xor 16, 32), and the probabilities feed the second MFMA straight from registers (never through LDS or
// HBM).  The MFMA k-slot <-> column mapping (slot (g,e): column blk*32 + (e<4 ? g*4+e : 16+g*4+e-4)) is applied
// identically to the register operand and to the transposing LDS read, which is all a contraction needs.
//
//   fwd   : online softmax over key tiles: S = QK^T*scale -> P (+dropout) -> O += P V      saves LSE[b,h,q]
//   bwd_dq: recompute P from LSE; dP = dO V^T; dS = P*(dP - delta)*scale; dQ += dS K       writes delta[b,h,q]
//   bwd_kv: (rows = keys, streams query tiles) recompute P^T; dV += Pdrop^T dO; dK += dS^T Q
// The backward recomputes the scores in both kernels (7 matmul units instead of 5) in exchange for no [B,h,T,T] tensor
// in HBM at all; attention is 3 % of the step's FLOPs.  Waves whose 16 rows lie entirely beyond T skip the arithmetic
// (T = 149: the third 64-row block has two idle waves) but keep loading tiles and taking the barriers.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int HD = 64;  // head dim

// 16-row periodic: all fragments of an image share one per-lane swizzle, so fragment addresses are
// lane base + compile-time constant (ds_read offset immediates instead of one address VGPR each)
// dropout counter of attention probability (bh, q, key): rows are padded to an even length so that keys 2j, 2j+1
// of a row always share one hash (common.h rng_pair); every fused kernel (forward, both backward kernels) uses it
__device__ __forceinline__ uint64_t attn_drop_idx(int64_t bh, int q, int key, int Tn) {
  return (uint64_t)(bh * Tn + q) * (uint64_t)((Tn + 1) & ~1) + (uint64_t)key;
}
// pair index of key 0 of query row q (attn_drop_idx(bh, q, 0, Tn) >> 1): a per-lane constant of the forward and dQ
// kernels, so a hash costs one 64-bit add instead of one 64-bit multiply
__device__ __forceinline__ uint64_t attn_row_pair0(int64_t bh, int q, int Tn) {
  return (uint64_t)(bh * Tn + q) * (uint64_t)(((Tn + 1) & ~1) >> 1);
}
// The pair index is 64-bit in general; when the whole launch has fewer than 2^32 pairs (B * heads * T * padded T / 2: every
// training shape) the host picks the 32-bit instantiation: the high word is zero, rng_pair's x = lo ^ key ^ umul24(hi, C)
// loses its third term and the per-hash 64-bit add becomes one v_add_u32 -- the SAME bits, three VALU slots fewer per hash.
__device__ __forceinline__ uint32_t rng_pair_i(uint32_t key, uint64_t pair_idx) { return rng_pair(key, pair_idx); }
__device__ __forceinline__ uint32_t rng_pair_i(uint32_t key, uint32_t pair_idx) {
  uint32_t x = pair_idx ^ key;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// keep-scales of 4 consecutive keys of a query row: two hashes.  row0 = pair index of the row's key k0 (k0 % 4 == 0,
// lane-dependent), key0_rel = distance of the first of the 4 keys from k0, a wave-uniform multiple of 4 -> one add
template <typename IDX>
__device__ __forceinline__ void attn_drop4_keys(uint32_t k, uint32_t thr, IDX row0, int key0_rel, float inv_keep,
                                                float (&ms)[4]) {
  const IDX pi = row0 + (IDX)(uint32_t)(key0_rel >> 1);
  const uint32_t h0 = rng_pair_i(k, pi), h1 = rng_pair_i(k, (IDX)(pi + 1));
  ms[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[1] = (h0 >> 16) >= thr ? inv_keep : 0.f;
  ms[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[3] = (h1 >> 16) >= thr ? inv_keep : 0.f;
}
// keep-scales of key column `key` (parity == lane parity) for 4 consecutive query rows q0 .. q0+3: the two lanes
// of a key pair hash two rows each and swap (one hash serves keys 2j and 2j+1 of a row)
// col0 = pair index of (query row 0 of this (utterance, head), this lane's key) = attn_drop_idx(bh, 0, key, Tn) >> 1, a
// per-lane constant; a row adds q * Tp2 (Tp2 = padded row length / 2: 24-bit operands, full-rate multiply)
template <typename IDX>
__device__ __forceinline__ void attn_drop4_rows(uint32_t k, uint32_t thr, IDX col0, uint32_t Tp2, int q0, int odd,
                                                float inv_keep, float (&ms)[4]) {
  const uint32_t qa = (uint32_t)(q0 + 2 * odd);
  const uint32_t h0 = rng_pair_i(k, (IDX)(col0 + (IDX)__umul24(qa, Tp2)));
  const uint32_t h1 = rng_pair_i(k, (IDX)(col0 + (IDX)__umul24(qa + 1u, Tp2)));
  const uint32_t p0 = __shfl_xor(h0, 1, 64), p1 = __shfl_xor(h1, 1, 64);
  const uint32_t r0 = odd ? p0 : h0, r1 = odd ? p1 : h1, r2 = odd ? h0 : p0, r3 = odd ? h1 : p1;
  const int sh = odd * 16;
  ms[0] = ((r0 >> sh) & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[1] = ((r1 >> sh) & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[2] = ((r2 >> sh) & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[3] = ((r3 >> sh) & 0xffffu) >= thr ? inv_keep : 0.f;
}
__device__ __forceinline__ int aswz(int row) { return ((row ^ (row >> 1)) & 3) | (row & 4); }   // measured map, see gemm.hip swz()

__device__ __forceinline__ frag8_t as_frag(uint4 v) {
  union { uint4 u; frag8_t f; } c;
  c.u = v;
  return c.f;
}
template <typename TE>
__device__ __forceinline__ frag8_t pack_frag(const float a[4], const float b[4]) {
  uint4 v;
  v.x = pack2<TE>(a[0], a[1]);
  v.y = pack2<TE>(a[2], a[3]);
  v.z = pack2<TE>(b[0], b[1]);
  v.w = pack2<TE>(b[2], b[3]);
  return as_frag(v);
}
// sum_e a[e] * b[e] over the 8 elements of two fragments (f32)
template <typename TE>
__device__ __forceinline__ float frag_dot(frag8_t a, frag8_t b) {
  union { frag8_t f; uint32_t w[4]; } ua, ub;
  ua.f = a;
  ub.f = b;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a0, a1, b0, b1;
    unpack2<TE>(ua.w[i], a0, a1);
    unpack2<TE>(ub.w[i], b0, b1);
    s = fmaf(a0, b0, s);
    s = fmaf(a1, b1, s);
  }
  return s;
}

// MFMA fragment (16 rows x 32 k) from the swizzled row-major image
__device__ __forceinline__ frag8_t lds_frag(const bf16_t* lds, int row0, int kk, int lane) {
  const int fr = lane & 15;                         // row0 is a multiple of 16: swizzle depends on fr only
  const int lane_off = fr * 64 + (((kk * 4 + (lane >> 4)) ^ aswz(fr)) << 3);
  return as_frag(*reinterpret_cast<const uint4*>(lds + lane_off + row0 * 64));
}
// the wave's own 16 rows straight from global into registers (2 k-steps)
__device__ __forceinline__ void reg_frag(frag8_t f[2], const bf16_t* g, int64_t gs, int row, int n_valid, int lane) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n_valid) v = *reinterpret_cast<const uint4*>(g + (int64_t)row * gs + kk * 32 + (lane >> 4) * 8);
    f[kk] = as_frag(v);
  }
}
// reductions over the four lanes {c, c+16, c+32, c+48} that share a row: gfx950 row / half swaps in the VALU
// (v_permlane16_swap: rows 1 <-> 0 and 3 <-> 2 of the two operands; v_permlane32_swap: upper half <-> lower half)
// instead of two ds_bpermute round trips with their per-call lane-index arithmetic
__device__ __forceinline__ float quad_max(float v) {
  const uint32_t u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t w = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
  const uint32_t u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t w = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// One output row (64 columns) is held by the four lanes g = 0..3 of a row: lane g has columns df * 16 + g * 4 .. + 3 of
// each 16-column block df -- four 8-byte pieces.  Lanes g and g ^ 1 exchange one piece per block pair
// (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second): the even lane then owns columns
// g * 4 .. + 7 of block 2p, the odd lane columns (g - 1) * 4 .. + 7 of block 2p + 1 -- two 16-byte write-through stores
// per lane instead of four 8-byte write-back ones.  (A kernel that leaves tens of MB of dirty lines in L2 pays ~2.4 us of
// write-back at its end, tools/probes/launch_gap_probe.hip; narrower sc1 stores are one fabric write each.)
// Every lane must call this (the swap is a cross-lane operation); `valid` guards the stores.
template <typename TE>
__device__ __forceinline__ void store_row4x4(bf16_t* row, int g, bool valid, const f32x4 (&o)[4]) {
  const bool odd = (g & 1) != 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const uint32_t ax = pack2<TE>(o[2 * p][0], o[2 * p][1]), ay = pack2<TE>(o[2 * p][2], o[2 * p][3]);
    const uint32_t bx = pack2<TE>(o[2 * p + 1][0], o[2 * p + 1][1]), by = pack2<TE>(o[2 * p + 1][2], o[2 * p + 1][3]);
    const auto sx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
    const auto sy = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
    // even lane: {own A, partner's A} = block 2p, 8 columns from g * 4; odd lane: {partner's B, own B} = block 2p + 1
    const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    if (valid) store16_wt(row + (2 * p + (odd ? 1 : 0)) * 16 + (g & 2) * 4, w);
  }
}

// ------------------------------------------------------------------------------------- transposing fragment reads
// The products that contract over the streamed axis (P V, dS K, P^T dO, dS^T Q) need their LDS operand transposed.
// The images stay ROW-MAJOR and the operand comes from the hardware transpose read ds_read_b64_tr_b16 (16 lanes x 8 B =
// a 4(row) x 16(col) block, lane i receives column i): no transposed copies in LDS.
typedef __attribute__((ext_vector_type(4))) short short4v_t;
typedef __attribute__((address_space(3))) short4v_t lds_s4v_t;

// MFMA operand fragment "columns d0..d0+15 x k-slots of 32-row block blk" from a row-major swizzled image.
// The swizzle of the rows a lane touches depends on the lane only (16-row periodic, blocks of 32 rows), so the
// address is  lane_off[d0/16] + blk * 2048 (+ 1024 for the second transposing read): four per-lane offsets
// computed once, everything else a ds_read immediate.
struct TrOff { int o[4]; };
__device__ __forceinline__ TrOff tr_offsets(int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int r = g * 4 + (i >> 2);                    // row within the 32-row block (first read; second = +16)
  const int sw = aswz(r);
  TrOff t;
#pragma unroll
  for (int df = 0; df < 4; ++df) t.o[df] = r * 64 + (((2 * df + ((i & 3) >> 1)) ^ sw) << 3) + ((i & 1) << 2);
  return t;
}
__device__ __forceinline__ frag8_t lds_frag_tr(const bf16_t* img, const TrOff& t, int df, int blk) {
  const bf16_t* p0 = img + t.o[df] + blk * 2048;
  const short4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_t*)p0);
  const short4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_t*)(p0 + 1024));
  union { struct { short4v_t a, b; } s; frag8_t v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// ===================================================================================== tiled kernels: any T
// Sequences longer than one workgroup's LDS images (T > 160: the paired-input model at T = 2*149+3 = 301, the 5 s
// clips of the large model at T = 249, evaluation utterances of up to ~145 s = 7249 frames) stream the OTHER
// sequence axis through LDS in tiles of 64 rows (double-buffered, next tile's global loads in flight under the
// current tile's MFMAs): the forward is an online-softmax (flash) loop over key tiles, the backward two kernels that
// recompute the probabilities from the saved log-sum-exp -- dQ (+ delta) over key tiles, dK / dV over query tiles --
// with the same per-fragment arithmetic, dropout stream and k-slot mapping as the single-workgroup kernels above.
// No [B, h, T, T] tensor exists anywhere; work per launch is O(T^2 d), LDS is 32-33 KiB whatever T is.
// Two geometries of the same kernels (template parameters NW = waves per workgroup, KT = rows of a streamed tile):
//   <4, 64>: 64 rows per workgroup, 64-row tiles (rounds 2-5).  At T = 149 a (utterance, head) is three workgroups, the last
//            one with two idle waves, and 2376 workgroups on 1024 slots (four per CU) are 2.32 lifetimes: the counters of
//            round 5 (profiles/r05_attention_pmc.txt) had the wave slots ~70 % occupied with the VALU ~90 % busy while resident.
//   <2, 32>: 32 rows per workgroup, 32-row tiles (round 6).  T = 149 is five workgroups with every wave active and five
//            tiles with no empty 16-key block; eight workgroups per CU (16 KiB of LDS each) = 3960 items on 2048 slots =
//            1.93 lifetimes, and the hardware dispatcher back-fills at twice the granularity.  Costs: every tile is loaded
//            for half as many rows (L2 -> LDS traffic x2, it was not the bound) and the online-softmax bookkeeping of the
//            forward (running max / rescale) is paid per 32 keys instead of 64.
// Both stage a tile as 2 x 16 B per thread; the dropout stream is indexed by absolute (query, key), so the two
// geometries draw the same mask.
struct TileRegs { uint4 v[2]; };
// rows row0 .. row0+KT-1 (x 64 elements) of a [*, 64]-column slab with row stride gs -> registers (2 x 16 B per thread)
template <int NT>
__device__ __forceinline__ void tile_load(TileRegs& r, const bf16_t* g, int64_t gs, int row0, int n_valid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + NT * i;
    const int row = row0 + (c >> 3), ch = c & 7;
    r.v[i] = make_uint4(0, 0, 0, 0);
    if (row < n_valid) r.v[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * gs + ch * 8);
  }
}
template <int NT>
__device__ __forceinline__ void tile_store(const TileRegs& r, bf16_t* lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + NT * i;
    const int row = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(lds + row * 64 + ((ch ^ aswz(row)) << 3)) = r.v[i];
  }
}

// The same tile straight from global memory into LDS (global_load_lds_dwordx4: no staging registers).  A wave-instruction
// moves 64 x 16 B = 8 rows of the image; lane l lands at chunk (l & 7) of row 8 p + (l >> 3), so the XOR swizzle is
// applied on the SOURCE side (the lane fetches chunk (l & 7) ^ aswz(row)).  The DMA cannot write zeros: rows past
// n_valid are CLAMPED onto the last valid row (finite data) and the caller makes their contribution vanish
// (dK / dV kernel: lse = +inf for those query rows -> p = 0).  KT / (8 NW) instructions per wave and operand.
typedef __attribute__((address_space(1))) const void attn_gvoid_t;
typedef __attribute__((address_space(3))) void attn_lvoid_t;
template <int NW, int KT>
__device__ __forceinline__ void tile_dma(const bf16_t* g, int64_t gs, int row0, int n_valid, bf16_t* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < KT / (8 * NW); ++i) {
    const int piece = wave * (KT / (8 * NW)) + i;
    const int row = piece * 8 + (lane >> 3);
    const int src_row = min(row0 + row, n_valid - 1);
    const int ch = (lane & 7) ^ aswz(row);
    __builtin_amdgcn_global_load_lds((attn_gvoid_t*)(g + (int64_t)src_row * gs + ch * 8), (attn_lvoid_t*)(lds + piece * 8 * 64), 16, 0, 0);
  }
}

// Workgroup -> (row tile, head, utterance).  The grid is ONE-dimensional and the index is remapped so that every XCD
// (workgroup id % 8) owns a contiguous run of logical indices: the ceil(T / 64) row tiles of one (utterance, head) --
// which all stream the SAME K / V (forward, dQ) or Q / dO (dK, dV) rows -- then sit on the same XCD at the same time and
// share them in its L2.  With the 3-D grid (row tile fastest) they were dealt out to three different XCDs and each
// fetched the operands from HBM itself: 121 / 152 / 155 MB per launch against ~60 algorithmic at T = 149 (PMC, round 2).
struct AttnBlock { int tile, h, b; };
__device__ __forceinline__ AttnBlock attn_block(int ntile_rows, int heads) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  if (heads < 0) { logical = bid; heads = -heads; }       // A/B switch (W2V2_ATTN_NO_XCD_REMAP): dispatch order
  AttnBlock o;
  o.tile = logical % ntile_rows;
  const int bh = logical / ntile_rows;
  o.h = bh % heads;
  o.b = bh / heads;
  return o;
}

template <typename TE, int NW, int KT, typename IDX>
__global__ __launch_bounds__(64 * NW, 4) void attn_fwd_tiled_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                             float* __restrict__ lse, int Tn, int heads_s, float scale,
                                                             float dp, float inv_keep, uint64_t seed) {
  static_assert(KT * 8 == 2 * 64 * NW, "a tile is staged as two 16-byte chunks per thread");
  constexpr int NT = 64 * NW, RW = 16 * NW, AT_TILE = KT;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][AT_TILE * 64];
  const AttnBlock blk = attn_block((Tn + RW - 1) / RW, heads_s);
  const int heads = heads_s < 0 ? -heads_s : heads_s;
  const int b = blk.b, h = blk.h, row_tile = blk.tile;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int q = row_tile * RW + wave * 16 + (lane & 15);
  const int64_t bh = (int64_t)b * heads + h;
  frag8_t qf[2];
  reg_frag(qf, qb, gs, q, Tn, lane);
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rk, rv;
  tile_load<NT>(rk, qb + H, gs, 0, Tn);
  tile_load<NT>(rv, qb + 2 * H, gs, 0, Tn);
  tile_store<NT>(rk, Ks[0]);
  tile_store<NT>(rv, Vs[0]);
  __syncthreads();
  float m = -INFINITY, l = 0.f;                       // running maximum in the log2 domain
  const float scale2 = scale * 1.4426950408889634f;
  const uint32_t thr = drop_thr16(dp);
  const uint32_t rkey = rng_key(seed);
  const IDX row0 = (IDX)(attn_row_pair0(bh, q, Tn) + (uint64_t)(g * 2));     // + the lane's 4 keys of a 16-key block
  f32x4 o[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool active = row_tile * RW + wave * 16 < Tn;       // wave-uniform: any of this wave's 16 rows valid
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {                              // next tile's loads fly under this tile's arithmetic
      tile_load<NT>(rk, qb + H, gs, (t + 1) * AT_TILE, Tn);
      tile_load<NT>(rv, qb + 2 * H, gs, (t + 1) * AT_TILE, Tn);
    }
    // 16-key blocks of this tile that hold a valid key (uniform): the blocks past the end of the sequence are skipped
    // outright -- at T = 149 that is two of the twelve blocks a query tile walks
    const int nfj = min(AT_TILE / 16, (Tn - t * AT_TILE + 15) >> 4);
    if (active) {
      // One body, two instantiations: FULL = a tile with four valid 16-key blocks and no key past T (every tile but the
      // last): straight-line code, no per-block branches, no masks
      auto body = [&](auto full_c, auto drop_c) {
        constexpr bool FULL = decltype(full_c)::value, DROP = decltype(drop_c)::value;
        float s[AT_TILE / 16][4];
        float tmax = -INFINITY;
        // scores in the LOG2 domain (scale * log2(e) folded into one multiply, v_exp_f32 is 2^x)
#pragma unroll
        for (int fj = 0; fj < AT_TILE / 16; ++fj) {
          if (FULL || fj < nfj) {
            f32x4 acc = mfma16<TE>(lds_frag(Ks[cur], fj * 16, 0, lane), qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            acc = mfma16<TE>(lds_frag(Ks[cur], fj * 16, 1, lane), qf[1], acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[j] * scale2;
              if constexpr (!FULL) v = (t * AT_TILE + fj * 16 + g * 4 + j < Tn) ? v : -INFINITY;
              s[fj][j] = v;
              tmax = fmaxf(tmax, v);
            }
          }
        }
        tmax = quad_max(tmax);
        const float mn = fmaxf(m, tmax);                  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m - mn);   // first tile: 2^(-inf) = 0
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int fj = 0; fj < AT_TILE / 16; ++fj) {
          if (FULL || fj < nfj) {
            float ms[4];
            if constexpr (DROP) attn_drop4_keys(rkey, thr, row0, t * AT_TILE + fj * 16, inv_keep, ms);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = __builtin_amdgcn_exp2f(s[fj][j] - mn);
              psum += p;                                  // the normaliser sums the probabilities BEFORE dropout
              s[fj][j] = DROP ? p * ms[j] : p;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[fj][j] = 0.f;
          }
        }
        l = l * alpha + quad_sum(psum);
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] *= alpha;
#pragma unroll
        for (int kb = 0; kb < AT_TILE / 32; ++kb) {
          if (FULL || 2 * kb < nfj) {
            const frag8_t pf = pack_frag<TE>(s[2 * kb], s[2 * kb + 1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) o[df] = mfma16<TE>(lds_frag_tr(Vs[cur], troff, df, kb), pf, o[df]);
          }
        }
      };
      const bool full = (t + 1) * AT_TILE <= Tn;
      if (dp > 0.f) {
        if (full) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{});
      } else {
        if (full) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{});
      }
    }   // active
    if (t + 1 < ntile) {
      tile_store<NT>(rk, Ks[cur ^ 1]);                    // last read in iteration t-1, behind that iteration's barrier
      tile_store<NT>(rv, Vs[cur ^ 1]);
    }
    __syncthreads();
  }
  const float inv = q < Tn ? 1.0f / l : 0.f;
#pragma unroll
  for (int df = 0; df < 4; ++df) o[df] *= inv;
  store_row4x4<TE>(ctx + ((int64_t)b * Tn + q) * H + h * HD, g, q < Tn, o);
  if (q < Tn && g == 0) lse[bh * Tn + q] = m * 0.6931471805599453f + __logf(l);
}

// dQ (and delta[q] = sum_d dO O) for 64 queries per workgroup, streaming key tiles
template <typename TE, int NW, int KT, typename IDX>
__global__ __launch_bounds__(64 * NW, 4) void attn_bwd_dq_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ ctx,
                                                                const bf16_t* __restrict__ dctx,
                                                                const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                                int Tn, int heads_s, float scale, float dp,
                                                                float inv_keep, uint64_t seed) {
  static_assert(KT * 8 == 2 * 64 * NW, "a tile is staged as two 16-byte chunks per thread");
  constexpr int NT = 64 * NW, RW = 16 * NW, AT_TILE = KT;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][AT_TILE * 64];
  const AttnBlock blk = attn_block((Tn + RW - 1) / RW, heads_s);
  const int heads = heads_s < 0 ? -heads_s : heads_s;
  const int b = blk.b, h = blk.h, row_tile = blk.tile;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const bf16_t* dob = dctx + (int64_t)b * Tn * H + h * HD;
  const bf16_t* ob = ctx + (int64_t)b * Tn * H + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int q = row_tile * RW + wave * 16 + (lane & 15);
  const int64_t bh = (int64_t)b * heads + h;
  frag8_t qf[2], dof[2], of[2];
  reg_frag(qf, qb, gs, q, Tn, lane);
  reg_frag(dof, dob, H, q, Tn, lane);
  reg_frag(of, ob, H, q, Tn, lane);
  float dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) dl += frag_dot<TE>(dof[kk], of[kk]);
  dl = quad_sum(dl);
  if (g == 0 && q < Tn) delta[bh * Tn + q] = dl;
  // P = exp(s * scale - lse) = 2^(s * scale2 - l2).  Rows beyond T need no mask: their q, dO are zero, so p = 1 and
  // dS = p * (0 - 0) = 0.  Keys beyond T (last tile only, uniform branch) keep theirs: p = exp(-lse) is unbounded
  // there and would meet a zero K row as inf * 0 after the 16-bit pack.
  const uint32_t thr = drop_thr16(dp);
  const uint32_t rkey = rng_key(seed);
  const IDX row0 = (IDX)(attn_row_pair0(bh, q, Tn) + (uint64_t)(g * 2));     // + the lane's 4 keys of a 16-key block
  const float l2 = q < Tn ? lse[bh * Tn + q] * 1.4426950408889634f : 0.f;
  const float scale2 = scale * 1.4426950408889634f;
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rk, rv;
  tile_load<NT>(rk, qb + H, gs, 0, Tn);
  tile_load<NT>(rv, qb + 2 * H, gs, 0, Tn);
  tile_store<NT>(rk, Ks[0]);
  tile_store<NT>(rv, Vs[0]);
  __syncthreads();
  f32x4 o[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool active = row_tile * RW + wave * 16 < Tn;
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {
      tile_load<NT>(rk, qb + H, gs, (t + 1) * AT_TILE, Tn);
      tile_load<NT>(rv, qb + 2 * H, gs, (t + 1) * AT_TILE, Tn);
    }
    const int nfj = min(AT_TILE / 16, (Tn - t * AT_TILE + 15) >> 4);      // 16-key blocks with a valid key (uniform)
    if (active)
#pragma unroll
    for (int kb = 0; kb < AT_TILE / 32; ++kb) {
      if (2 * kb < nfj) {
      float ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fj = 2 * kb + hf;
        if (fj < nfj) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            sa = mfma16<TE>(lds_frag(Ks[cur], fj * 16, kk, lane), qf[kk], sa);
            pa = mfma16<TE>(lds_frag(Vs[cur], fj * 16, kk, lane), dof[kk], pa);
          }
          const int key0 = t * AT_TILE + fj * 16 + g * 4;
          float ms[4] = {1.f, 1.f, 1.f, 1.f};
          if (dp > 0.f) attn_drop4_keys(rkey, thr, row0, t * AT_TILE + fj * 16, inv_keep, ms);
          if ((t + 1) * AT_TILE > Tn) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = key0 + j < Tn ? __builtin_amdgcn_exp2f(fmaf(sa[j], scale2, -l2)) : 0.f;
              ds2[hf][j] = p * (pa[j] * ms[j] - dl) * scale;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = __builtin_amdgcn_exp2f(fmaf(sa[j], scale2, -l2));
              ds2[hf][j] = p * (pa[j] * ms[j] - dl) * scale;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) ds2[hf][j] = 0.f;
        }
      }
      const frag8_t pf = pack_frag<TE>(ds2[0], ds2[1]);
#pragma unroll
      for (int df = 0; df < 4; ++df) o[df] = mfma16<TE>(lds_frag_tr(Ks[cur], troff, df, kb), pf, o[df]);
      }
    }
    if (t + 1 < ntile) {
      tile_store<NT>(rk, Ks[cur ^ 1]);
      tile_store<NT>(rv, Vs[cur ^ 1]);
    }
    __syncthreads();
  }
  store_row4x4<TE>(dqkv + ((int64_t)b * Tn + q) * gs + h * HD, g, q < Tn, o);
}

// dK, dV for 64 keys per workgroup, streaming query tiles (Q, dO, LSE, delta)
template <typename TE, int NW, int KT, typename IDX, bool DMA>
__global__ __launch_bounds__(64 * NW, DMA ? 4 : 3) void attn_bwd_kv_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ dctx,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int Tn, int heads_s,
                                                                float scale, float dp, float inv_keep, uint64_t seed) {
  static_assert(KT * 8 == 2 * 64 * NW, "a tile is staged as two 16-byte chunks per thread");
  constexpr int NT = 64 * NW, RW = 16 * NW, AT_TILE = KT;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Os[2][AT_TILE * 64];      // dO
  __shared__ __attribute__((aligned(16))) float lse_s[2][AT_TILE];
  __shared__ __attribute__((aligned(16))) float del_s[2][AT_TILE];
  const AttnBlock blk = attn_block((Tn + RW - 1) / RW, heads_s);
  const int heads = heads_s < 0 ? -heads_s : heads_s;
  const int b = blk.b, h = blk.h, row_tile = blk.tile;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const bf16_t* dob = dctx + (int64_t)b * Tn * H + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int key = row_tile * RW + wave * 16 + (lane & 15);
  const int64_t bh = (int64_t)b * heads + h;
  frag8_t kf[2], vf[2];
  reg_frag(kf, qb + H, gs, key, Tn, lane);
  reg_frag(vf, qb + 2 * H, gs, key, Tn, lane);
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rq, ro;
  float rl = 0.f, rd = 0.f;
  // (DMA: query rows past T arrive as copies of row T - 1; +inf in the log2-domain LSE makes their p exactly 0)
  auto row_load = [&](int t) {
    if (threadIdx.x < AT_TILE) {
      const int r = t * AT_TILE + threadIdx.x;
      rl = r < Tn ? lse[bh * Tn + r] * 1.4426950408889634f : (DMA ? INFINITY : 0.f);
      rd = r < Tn ? delta[bh * Tn + r] : 0.f;
    }
  };
  auto row_store = [&](int buf) {
    if (threadIdx.x < AT_TILE) { lse_s[buf][threadIdx.x] = rl; del_s[buf][threadIdx.x] = rd; }
  };
  if constexpr (DMA) {
    tile_dma<NW, KT>(qb, gs, 0, Tn, Qs[0], wave, lane);
    tile_dma<NW, KT>(dob, H, 0, Tn, Os[0], wave, lane);
    row_load(0);
    row_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    tile_load<NT>(rq, qb, gs, 0, Tn);
    tile_load<NT>(ro, dob, H, 0, Tn);
    row_load(0);
    tile_store<NT>(rq, Qs[0]);
    tile_store<NT>(ro, Os[0]);
    row_store(0);
  }
  __syncthreads();
  f32x4 dv[4], dk[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) { dv[df] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[df] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const bool active = row_tile * RW + wave * 16 < Tn;
  const uint32_t rkey = rng_key(seed), thr = drop_thr16(dp);
  const IDX col0 = (IDX)(attn_drop_idx(bh, 0, key, Tn) >> 1);
  const uint32_t Tp2 = (uint32_t)(((Tn + 1) & ~1) >> 1);
  const float scale2 = scale * 1.4426950408889634f;
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {
      if constexpr (DMA) {        // buffer cur ^ 1 was last read in iteration t - 1, behind that iteration's barrier
        tile_dma<NW, KT>(qb, gs, (t + 1) * AT_TILE, Tn, Qs[cur ^ 1], wave, lane);
        tile_dma<NW, KT>(dob, H, (t + 1) * AT_TILE, Tn, Os[cur ^ 1], wave, lane);
      } else {
        tile_load<NT>(rq, qb, gs, (t + 1) * AT_TILE, Tn);
        tile_load<NT>(ro, dob, H, (t + 1) * AT_TILE, Tn);
      }
      row_load(t + 1);
    }
    const int nfq = min(AT_TILE / 16, (Tn - t * AT_TILE + 15) >> 4);      // 16-query blocks with a valid row (uniform)
    if (active)
#pragma unroll
    for (int qb2 = 0; qb2 < AT_TILE / 32; ++qb2) {
      if (2 * qb2 < nfq) {
      float pt2[2][4], ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fq = 2 * qb2 + hf;
        if (fq < nfq) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            sa = mfma16<TE>(lds_frag(Qs[cur], fq * 16, kk, lane), kf[kk], sa);
            pa = mfma16<TE>(lds_frag(Os[cur], fq * 16, kk, lane), vf[kk], pa);
          }
          const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[cur][fq * 16 + g * 4]);
          const float4 d4 = *reinterpret_cast<const float4*>(&del_s[cur][fq * 16 + g * 4]);
          const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
          const int q0 = t * AT_TILE + fq * 16 + g * 4;
          float ms[4] = {1.f, 1.f, 1.f, 1.f};
          if (dp > 0.f) attn_drop4_rows(rkey, thr, col0, Tp2, q0, key & 1, inv_keep, ms);
          // log2 domain (lse_s holds lse * log2 e).  Query rows beyond T need no mask (their q and dO rows are zero: p = 1
          // meets dO = 0 and dP - delta = 0); key columns beyond T keep theirs (p = 2^-lse is unbounded there)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float p = key < Tn ? __builtin_amdgcn_exp2f(fmaf(sa[j], scale2, -la[j])) : 0.f;
            pt2[hf][j] = p * ms[j];
            ds2[hf][j] = p * (pa[j] * ms[j] - da[j]) * scale;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { pt2[hf][j] = 0.f; ds2[hf][j] = 0.f; }
        }
      }
      const frag8_t pf = pack_frag<TE>(pt2[0], pt2[1]);
      const frag8_t sf = pack_frag<TE>(ds2[0], ds2[1]);
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        dv[df] = mfma16<TE>(lds_frag_tr(Os[cur], troff, df, qb2), pf, dv[df]);
        dk[df] = mfma16<TE>(lds_frag_tr(Qs[cur], troff, df, qb2), sf, dk[df]);
      }
      }
    }
    if (t + 1 < ntile) {
      if constexpr (!DMA) {
        tile_store<NT>(rq, Qs[cur ^ 1]);
        tile_store<NT>(ro, Os[cur ^ 1]);
      }
      row_store(cur ^ 1);
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  bf16_t* dstk = dqkv + ((int64_t)b * Tn + key) * gs + H + h * HD;
  store_row4x4<TE>(dstk, g, key < Tn, dk);
  store_row4x4<TE>(dstk + H, g, key < Tn, dv);
}

// ------------------------------------------------------------------------------------- host
static int attn_check(const char* nm, int B, int T, int heads, int d, int dtype, float drop_p) {
  W2V2_REQUIRE(B > 0 && T > 0 && heads > 0, "%s: bad shape", nm);
  W2V2_REQUIRE(d == HD, "%s: fused attention needs head dim 64 (got %d); use the unfused path", nm, d);
  W2V2_REQUIRE(dtype == W2V2_BF16 || dtype == W2V2_F16,
               "%s: fused attention needs 16-bit activations; the f32 parity mode uses the unfused path", nm);
  W2V2_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: bad dropout p", nm);
  return 0;
}

static const bool g_attn_no_remap = getenv("W2V2_ATTN_NO_XCD_REMAP") != nullptr;
// geometry switch (see the tile helpers): 0 / unset = by sequence length, 64 = force <4, 64>, 32 = force <2, 32>
// (read on every call, ~0.1 us: a test can compare the two geometries inside one process)
static int attn_geom_env() { const char* e = getenv("W2V2_ATTN_GEOM"); return e ? atoi(e) : 0; }
// The small geometry is taken when it pads the sequence to FEWER rows than the large one (T = 149: 160 instead of 192,
// T = 199: 224 instead of 256) and the sequence is short; where both pad alike (T = 249, 301) the large one wins by its
// K / V reuse (measured, tools/attn_bench.py: T = 149 fwd 28.1 -> 26.6 us, bwd 71.1 -> 65.5; T = 249 fwd 49.5 -> 53.3).
// W2V2_ATTN_IDX64 forces the 64-bit dropout counters (tests compare the two index widths).
constexpr int ATTN_SMALL_GEOM_MAX_T = 512;
static bool attn_small_geom(int T) {
  const int g = attn_geom_env();
  return g == 32 || (g != 64 && T <= ATTN_SMALL_GEOM_MAX_T && ((T + 31) & ~31) < ((T + 63) & ~63));
}

// true when every pair index of the launch fits 32 bits (see rng_pair_i)
static bool attn_idx32(int B, int T, int heads) {
  if (getenv("W2V2_ATTN_IDX64")) return false;
  return (uint64_t)B * heads * T * (uint64_t)(((T + 1) & ~1) >> 1) < (1ull << 32);
}

template <typename TE, int NW, int KT, typename IDX>
static int attention_fwd_g(const void* qkv, void* ctx, float* lse, int B, int T, int heads, float scale, float drop_p,
                           uint64_t seed, void* stream) {
  const float ik = 1.0f / (1.0f - drop_p);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)(cdiv(T, 16 * NW) * heads * B));       // see attn_block
  hipLaunchKernelGGL((attn_fwd_tiled_kernel<TE, NW, KT, IDX>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv, (bf16_t*)ctx, lse, T,
                     g_attn_no_remap ? -heads : heads, scale, drop_p, ik, seed);
  W2V2_CHECK_LAUNCH("attention_fwd");
  return 0;
}
template <typename TE>
static int attention_fwd_t(const void* qkv, void* ctx, float* lse, int B, int T, int heads, float scale, float drop_p,
                           uint64_t seed, void* stream) {
  const bool i32 = attn_idx32(B, T, heads);
  if (attn_small_geom(T))
    return i32 ? attention_fwd_g<TE, 2, 32, uint32_t>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream)
               : attention_fwd_g<TE, 2, 32, uint64_t>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream);
  return i32 ? attention_fwd_g<TE, 4, 64, uint32_t>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream)
             : attention_fwd_g<TE, 4, 64, uint64_t>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream);
}

extern "C" int w2v2_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int heads, int d, float scale,
                                  float drop_p, uint64_t seed, int dtype, void* stream) {
  if (attn_check("attention_fwd", B, T, heads, d, dtype, drop_p)) return -1;
  W2V2_REQUIRE(qkv && ctx && lse, "attention_fwd: null pointer");
  W2V2_DISPATCH_16(dtype, "attention_fwd",
                   return attention_fwd_t<AT>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream););
  return 0;
}

template <typename TE, int NW, int KT, typename IDX>
static int attention_bwd_g(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                           float* delta, int B, int T, int heads, float scale, float drop_p, uint64_t seed,
                           void* stream) {
  const float ik = 1.0f / (1.0f - drop_p);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)(cdiv(T, 16 * NW) * heads * B));       // see attn_block
  hipLaunchKernelGGL((attn_bwd_dq_tiled_kernel<TE, NW, KT, IDX>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv, (const bf16_t*)ctx,
                     (const bf16_t*)dctx, lse, (bf16_t*)dqkv, delta, T, g_attn_no_remap ? -heads : heads, scale, drop_p, ik,
                     seed);
  // dK / dV with LDS-DMA tile loads (no staging registers: 128 VGPRs, four waves per SIMD); W2V2_ATTN_KV_NO_DMA: A/B switch
  if (NW == 2 && getenv("W2V2_ATTN_KV_NO_DMA") == nullptr)
    hipLaunchKernelGGL((attn_bwd_kv_tiled_kernel<TE, NW, KT, IDX, NW == 2>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv,
                       (const bf16_t*)dctx, lse, (const float*)delta, (bf16_t*)dqkv, T, g_attn_no_remap ? -heads : heads, scale,
                       drop_p, ik, seed);
  else
    hipLaunchKernelGGL((attn_bwd_kv_tiled_kernel<TE, NW, KT, IDX, false>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv,
                       (const bf16_t*)dctx, lse, (const float*)delta, (bf16_t*)dqkv, T, g_attn_no_remap ? -heads : heads, scale,
                       drop_p, ik, seed);
  W2V2_CHECK_LAUNCH("attention_bwd");
  return 0;
}
template <typename TE>
static int attention_bwd_t(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                           float* delta, int B, int T, int heads, float scale, float drop_p, uint64_t seed,
                           void* stream) {
  const bool i32 = attn_idx32(B, T, heads);
  if (attn_small_geom(T))
    return i32 ? attention_bwd_g<TE, 2, 32, uint32_t>(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, scale, drop_p, seed, stream)
               : attention_bwd_g<TE, 2, 32, uint64_t>(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, scale, drop_p, seed, stream);
  return i32 ? attention_bwd_g<TE, 4, 64, uint32_t>(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, scale, drop_p, seed, stream)
             : attention_bwd_g<TE, 4, 64, uint64_t>(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, scale, drop_p, seed, stream);
}

extern "C" int w2v2_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                                  float* delta, int B, int T, int heads, int d, float scale, float drop_p,
                                  uint64_t seed, int dtype, void* stream) {
  if (attn_check("attention_bwd", B, T, heads, d, dtype, drop_p)) return -1;
  W2V2_REQUIRE(qkv && ctx && dctx && lse && dqkv && delta, "attention_bwd: null pointer");
  W2V2_DISPATCH_16(dtype, "attention_bwd",
                   return attention_bwd_t<AT>(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, scale, drop_p, seed, stream););
  return 0;
}
