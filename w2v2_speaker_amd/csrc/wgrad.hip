// wgrad.hip -- grouped weight-gradient GEMM:  dW_p[o][i] = sum_t dY_p[t][o] * X_p[t][i]   (+ dbias_p[o] = sum_t dY_p[t][o])
// for up to 12 (dY, X) pairs in ONE launch (the four Linear layers of a transformer block: QKV, out-proj,
// FFN1, FFN2 -> 432 full 128x128 tiles at w2v2-base, enough to fill 256 CUs without split-K, so there
// are no atomics and the gradients are bitwise reproducible).
//
// Both operands are K-major here (K = tokens is the OUTER index of the activation matrices).  They are
// staged HBM -> LDS with global_load_lds_dwordx4 in their natural [k][m] layout (a wave-instruction
// moves 4 token rows x 256 B) and the MFMA fragments are produced by the hardware transpose read
// ds_read_b64_tr_b16 (16 lanes x 8 B: a 4(k) x 16(m) block, lane i receives column i), so nothing is
// ever transposed in HBM or in registers.  The 32-byte segments of a token row are XOR-swizzled with
// f(k) = (k & 3) | ((k >> 1) & 4) -- on the DMA source address and again on the read -- which spreads
// the 8 rows a half-wave touches over all 8 segments (conflict-free ds_read_b64_tr_b16).
// Contract: token rows [tokens, tokens_padded) of every operand are readable and ZERO
// (tokens_padded = tokens rounded up to 64); n_out, n_in multiples of 8; 16-byte aligned rows.
#include "wgrad_common.h"

__device__ __forceinline__ int wg_f(int r) { return (r & 3) | ((r >> 1) & 4); }

__device__ __forceinline__ frag8_t tr_frag(const bf16_t* tile, int kk, int seg, int lane) {
  // k-slots of lane group g: rows kk*32 + g*8 + {0..7}; two 4x16 transposing reads
  const int i = lane & 15, g = lane >> 4;
  const int r = kk * 32 + g * 8 + (i >> 2);
  const int f = (i >> 2) | ((g & 1) << 2);
  const bf16_t* p = tile + r * 128 + ((seg ^ f) << 4) + ((i & 3) << 2);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)p);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(p + 4 * 128));
  union { struct { short4v a, b; } s; frag8_t v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

template <typename TE>
__global__ __launch_bounds__(256) void wgrad_grouped_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);   // [2][64 k][128 m]
  bf16_t* Bs = As + 2 * 64 * 128;                     // [2][64 k][128 n]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // tile -> (problem, tm, tn); XCD-aware order as in gemm.hip
  int tile;
  {
    const int nwg = a.total_tiles, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < a.n_problems && tile >= a.p[i].tile_begin) pi = i;
  const WgProblem& P = a.p[pi];
  const int t = tile - P.tile_begin;
  const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;

  // per-lane DMA sources: piece j of a wave covers token rows (wave*4 + j)*4 .. +3 of the 64-row tile
  const int c16 = lane & 15, r4 = lane >> 4;
  const bf16_t* ap[4];
  const bf16_t* bp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 4 + r4;
    const int c = c16 ^ (wg_f(r) << 1);                    // logical 16-byte chunk held at physical chunk c16
    const int mc = min(m0 + c * 8, P.n_out - 8);           // clamp: columns beyond the matrix are never stored
    const int nc = min(n0 + c * 8, P.n_in - 8);
    ap[j] = P.dY + (int64_t)r * P.ld_dy + mc;
    bp[j] = P.X + (int64_t)r * P.ld_x + nc;
  }
  const int64_t astep = 64 * P.ld_dy, bstep = 64 * P.ld_x;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (P.dbias != nullptr) && tn == 0 && wn == 0;
  const uint32_t one2 = ones_pair<TE>();

  auto stage = [&](int buf, int kt) {
    bf16_t* ad = As + buf * 64 * 128 + wave * 16 * 128;
    bf16_t* bd = Bs + buf * 64 * 128 + wave * 16 * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * astep), (lvoid_t*)(ad + j * 4 * 128), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kt * bstep), (lvoid_t*)(bd + j * 4 * 128), 16, 0, 0);
  };
  auto compute = [&](int buf) {
    const bf16_t* Ac = As + buf * 64 * 128;
    const bf16_t* Bc = Bs + buf * 64 * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(Ac, kk, wm * 4 + i, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(Bc, kk, wn * 4 + j, lane);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          union { frag8_t v; uint32_t p[4]; } u;
          u.v = af[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[i] = pair_sum_add<TE>(u.p[e], one2, bsum[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<TE>(bfr[j], af[i], acc[i][j]);
    }
  };

  const int nk = a.ktiles;
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

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = bsum[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
      if ((lane >> 4) == 0 && m < P.n_out) P.dbias[m] = s;
    }
  }
  // coalesced f32 tile store through LDS (two 64-row halves, as in gemm.hip)
  float* stagef = reinterpret_cast<float*>(smem_raw);
  constexpr int PITCH = 128 + 4;
  const int frow = lane & 15, fk = lane >> 4;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(stagef + (i * 16 + frow) * PITCH + wn * 64 + j * 16 + fk * 4) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 256 * it;                 // 64 rows x 32 float4 chunks
      const int r = c >> 5, ch = c & 31;
      const int m = m0 + pass * 64 + r, n = n0 + ch * 4;
      if (m < P.n_out && n + 4 <= P.n_in)
        store16_wt(P.dW + (int64_t)m * P.ld_dw + n, *reinterpret_cast<const uint4*>(stagef + r * PITCH + ch * 4));
    }
  }
}

// ------------------------------------------------------------------------------ 256 x 128 ring kernel
// Same data path (LDS-DMA in natural [k][m] layout + ds_read_b64_tr_b16 fragments), but a 256(n_out) x
// 128(n_in) x 64(tokens) block tile for 8 waves (4 x 2, 64 x 64 per wave) on the 3-stage LDS ring of
// gemm16_ring_256x128_kernel: while tile t is multiplied, tiles t+1 and t+2 are in flight; per K tile one raw
// s_barrier and `s_waitcnt vmcnt(6)` (6 = this wave's DMA pieces per stage: 4 x [2 rows x 512 B] of dY,
// 2 x [4 rows x 256 B] of X).  The four Linear layers of a w2v2-base block are 216 tiles = one round on 256 CUs.
template <int S> __device__ __forceinline__ void wg_wait_vmcnt() {
  if constexpr (S == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

template <typename TE>
__global__ __launch_bounds__(512) void wgrad_grouped_ring_kernel(const WgArgs a) {
  constexpr int BM = 256, BN = 128;
  constexpr int STAGE = 64 * (BM + BN);             // elements per stage: A [64][256] then B [64][128]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 1, wn = wave & 1;

  int tile;
  {
    const int nwg = a.total_tiles, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < a.n_problems && tile >= a.p[i].tile_begin) pi = i;
  const WgProblem& P = a.p[pi];
  const int t = tile - P.tile_begin;
  const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // DMA sources.  A: piece j of a wave = token rows (wave*4 + j)*2 + {0,1}, 32 chunks of 16 B per row;
  //              B: piece j of a wave = token rows (wave*2 + j)*4 + {0..3}, 16 chunks per row.
  const bf16_t* ap[4];
  const bf16_t* bp[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 2 + (lane >> 5);
    const int c = (lane & 31) ^ (wg_f(r) << 1);
    const int mc = min(m0 + c * 8, P.n_out - 8);
    ap[j] = P.dY + (int64_t)r * P.ld_dy + mc;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (wg_f(r) << 1);
    const int nc = min(n0 + c * 8, P.n_in - 8);
    bp[j] = P.X + (int64_t)r * P.ld_x + nc;
  }
  const int64_t astep = 64 * P.ld_dy, bstep = 64 * P.ld_x;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dbias[m] = sum_t dY[t][m]: the waves of n-tile 0 / wn 0 add up the dY fragments they hold anyway, two bf16 per
  // v_dot2c_f32_bf16 against (1, 1) -- 4 VALU ops per fragment instead of 16 converts+adds
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (P.dbias != nullptr) && tn == 0 && wn == 0;
  const uint32_t one2 = ones_pair<TE>();

  auto stage = [&](bf16_t* base, int kt) {
    bf16_t* ad = base + wave * 8 * BM;
    bf16_t* bd = base + 64 * BM + wave * 8 * BN;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * astep), (lvoid_t*)(ad + j * 2 * BM), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kt * bstep), (lvoid_t*)(bd + j * 4 * BN), 16, 0, 0);
  };
  // per-lane fragment byte offsets: row (g*8 + i/4), 8-byte piece (i%4), physical segment (seg ^ f) with
  // seg = w*4 + x (x = 0..3 compile time) and f = (i/4) | ((g&1) << 2)  ->  ((w ^ f>>2) << 2) | (x ^ (f&3)).
  // The transposing reads are issued as inline asm: the compiler treats the ds_read_tr builtin as possibly
  // aliasing the LDS-DMA writes in flight and would put `s_waitcnt vmcnt(0)` in front of every group of
  // reads, draining the ring.  Which stage is being read vs written is guaranteed by the ring protocol.
  const int li = lane & 15, lg = lane >> 4;
  const int fr = lg * 8 + (li >> 2);
  const int flo = (li >> 2), fhi = lg & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    aoff[x] = lds0 + 2u * (fr * BM + (((((wm ^ fhi) & 3) << 2) | (x ^ flo)) << 4) + ((li & 3) << 2));
    boff[x] = lds0 + 2u * (64 * BM + fr * BN + (((((wn ^ fhi) & 1) << 2) | (x ^ flo)) << 4) + ((li & 3) << 2));
  }
  union Frag { struct { short4v a, b; } s; frag8_t v; };
#define W2V2_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr) : "memory")
#define W2V2_FRAG_A(f, addr, KK)                                   \
  if (KK == 0) { W2V2_TR_READ(f.s.a, addr, 0);     W2V2_TR_READ(f.s.b, addr, 2048); }  \
  else         { W2V2_TR_READ(f.s.a, addr, 16384); W2V2_TR_READ(f.s.b, addr, 18432); }
#define W2V2_FRAG_B(f, addr, KK)                                   \
  if (KK == 0) { W2V2_TR_READ(f.s.a, addr, 0);    W2V2_TR_READ(f.s.b, addr, 1024); }   \
  else         { W2V2_TR_READ(f.s.a, addr, 8192); W2V2_TR_READ(f.s.b, addr, 9216); }
  static_assert(4 * BM * 2 == 2048 && 32 * BM * 2 == 16384 && 4 * BN * 2 == 1024 && 32 * BN * 2 == 8192, "offsets");
  auto landed = [&](Frag (&f)[4]) {   // LDS returns in order: all reads issued so far have landed after this
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[i].s.a), "+v"(f[i].s.b));
  };
  auto mma = [&](Frag (&af)[4], Frag (&bfr)[4]) {
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        union { frag8_t v; uint32_t p[4]; } u;
        u.v = af[i].v;
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[i] = pair_sum_add<TE>(u.p[e], one2, bsum[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16<TE>(bfr[j].v, af[i].v, acc[i][j]);
  };
  auto compute = [&](uint32_t sbytes) {
    uint32_t aa[4], ba[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { aa[x] = aoff[x] + sbytes; ba[x] = boff[x] + sbytes; }
    Frag a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { W2V2_FRAG_A(a0[x], aa[x], 0) }
#pragma unroll
    for (int x = 0; x < 4; ++x) { W2V2_FRAG_B(b0[x], ba[x], 0) }
    landed(a0);
    landed(b0);
#pragma unroll
    for (int x = 0; x < 4; ++x) { W2V2_FRAG_A(a1[x], aa[x], 1) }
#pragma unroll
    for (int x = 0; x < 4; ++x) { W2V2_FRAG_B(b1[x], ba[x], 1) }
    __builtin_amdgcn_sched_barrier(0);   // k-step 1 reads are in flight under the MFMAs of k-step 0 ...
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);   // ... and their wait comes after those MFMAs
    landed(a1);
    landed(b1);
    mma(a1, b1);
  };
  const int nk = a.ktiles;
#define W2V2_WG_RING_STEP(cur, nxt)                                      \
  {                                                                      \
    if (kt + 1 < nk) wg_wait_vmcnt<6>(); else wg_wait_vmcnt<0>();        \
    __builtin_amdgcn_s_barrier();                                        \
    if (kt + 2 < nk) stage(smem + (nxt) * STAGE, kt + 2);                \
    compute((cur) * STAGE * 2u);                                         \
    ++kt;                                                                \
  }
  if (nk > 0) stage(smem, 0);
  if (nk > 1) stage(smem + STAGE, 1);
  int kt = 0;
  while (kt < nk) {
    W2V2_WG_RING_STEP(0, 2)
    if (kt >= nk) break;
    W2V2_WG_RING_STEP(1, 0)
    if (kt >= nk) break;
    W2V2_WG_RING_STEP(2, 1)
  }
#undef W2V2_WG_RING_STEP
#undef W2V2_FRAG_A
#undef W2V2_FRAG_B
#undef W2V2_TR_READ

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = bsum[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
      if ((lane >> 4) == 0 && m < P.n_out) P.dbias[m] = s;
    }
  }
  // coalesced f32 tile store through LDS: two passes of 128 rows
  float* stagef = reinterpret_cast<float*>(smem_raw);
  constexpr int PITCH = BN + 4;
  const int frow = lane & 15, fk = lane >> 4;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(stagef + ((wm & 1) * 64 + i * 16 + frow) * PITCH + wn * 64 + j * 16 + fk * 4) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 512 * it;                 // 128 rows x 32 float4 chunks
      const int r = c >> 5, ch = c & 31;
      const int m = m0 + pass * 128 + r, n = n0 + ch * 4;
      if (m < P.n_out && n + 4 <= P.n_in)
        store16_wt(P.dW + (int64_t)m * P.ld_dw + n, *reinterpret_cast<const uint4*>(stagef + r * PITCH + ch * 4));
    }
  }
}


// ------------------------------------------------------------------------------ 256 x 256 x 32, 4-stage ring
// Same idea as gemm16_ring_256x256_kernel: half the L2 -> LDS bytes per flop of the 256x128 tile.  A block of w2v2-base
// has only 108 such tiles, so the host groups the four Linear layers of TWO transformer blocks (8 problems = 216
// tiles, one round on 256 CUs): still no split-K, no atomics, bitwise reproducible.  8 waves as 2 (n_out) x 4 (n_in),
// 128 x 64 per wave; four 32 KiB stages [32 tokens][256 + 256] in natural K-major layout, three in flight
// (`s_waitcnt vmcnt(8)`, 4 DMA pieces per wave and stage); rolled ring loop (see gemm.hip on why).
// (Round 3 also carried a stream-K owner / helper form of this kernel for the 40 idle CUs of a 216-tile launch and a
// blocked tile order: four variants, all correct, none faster -- DESIGN.md section 4 keeps the measurements; deleted in
// round 4.)
template <typename TE>
__global__ __launch_bounds__(512) void wgrad_grouped_ring4_kernel(const WgArgs a) {
  constexpr int BM = 256, BN = 256, BK = 32;
  constexpr int STAGE = BK * (BM + BN);             // elements per stage: A [32][256] then B [32][256]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wm = wave >> 2, wn = wave & 3;
  const int nk_tile = a.ktiles * 2;                  // stages of 32 tokens per tile

  int tile;
  {
    const int nwg = a.total_tiles, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int nk = nk_tile;
  constexpr int k_first = 0;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < a.n_problems && tile >= a.p[i].tile_begin) pi = i;
  const WgProblem& P = a.p[pi];
  const int t = tile - P.tile_begin;
  const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // DMA pieces: piece j of a wave = token rows (wave*2 + j)*2 + {0,1} of the 32-row stage, 32 chunks of 16 B per row
  const bf16_t* ap[2];
  const bf16_t* bp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 2 + (lane >> 5);
    const int c = (lane & 31) ^ (wg_f(r) << 1);
    ap[j] = P.dY + (int64_t)(r + k_first * BK) * P.ld_dy + min(m0 + c * 8, P.n_out - 8);
    bp[j] = P.X + (int64_t)(r + k_first * BK) * P.ld_x + min(n0 + c * 8, P.n_in - 8);
  }
  const int64_t astep = (int64_t)BK * P.ld_dy, bstep = (int64_t)BK * P.ld_x;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (P.dbias != nullptr) && tn == 0 && wn == 0;
  const uint32_t one2 = ones_pair<TE>();

  auto stage = [&](int sidx, int kt) {
    bf16_t* ad = smem + sidx * STAGE + wave * 4 * BM;
    bf16_t* bd = smem + sidx * STAGE + BK * BM + wave * 4 * BN;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[j] + kt * astep), (lvoid_t*)(ad + j * 2 * BM), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[j] + kt * bstep), (lvoid_t*)(bd + j * 2 * BN), 16, 0, 0);
  };
  // fragment byte offsets (see wgrad_grouped_ring_kernel): row g*8 + i/4 (+4 for the second read), 8-byte piece
  // i%4, physical 32-byte segment (seg ^ f), f = (i/4) | ((g&1) << 2); seg = wm*8 + x (A), wn*4 + x (B)
  const int li = lane & 15, lg = lane >> 4;
  const int fr = lg * 8 + (li >> 2);
  const int ff = (li >> 2) | ((lg & 1) << 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  uint32_t aoff[8], boff[4];
#pragma unroll
  for (int x = 0; x < 8; ++x) aoff[x] = lds0 + 2u * (fr * BM + (((wm * 8 + x) ^ ff) << 4) + ((li & 3) << 2));
#pragma unroll
  for (int x = 0; x < 4; ++x) boff[x] = lds0 + 2u * (BK * BM + fr * BN + (((wn * 4 + x) ^ ff) << 4) + ((li & 3) << 2));
  union Frag { struct { short4v a, b; } s; frag8_t v; };
#define W2V2_TR4(f, addr)                                                                              \
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.s.a) : "v"(addr) : "memory");                      \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(f.s.b) : "v"(addr) : "memory")
  static_assert(4 * BM * 2 == 2048 && 4 * BN * 2 == 2048, "second transposing read = +4 token rows");

  __builtin_amdgcn_s_barrier();
  if (nk > 0) stage(0, 0);
  if (nk > 1) stage(1, 1);
  if (nk > 2) stage(2, 2);
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < nk) stage((kt + 3) & 3, kt + 3);
    const uint32_t sb = (uint32_t)(kt & 3) * (STAGE * 2u);
    Frag bf_[4], af[8];
#pragma unroll
    for (int x = 0; x < 4; ++x) { const uint32_t ad = boff[x] + sb; W2V2_TR4(bf_[x], ad); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { const uint32_t ad = aoff[x] + sb; W2V2_TR4(af[x], ad); }
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf_[x].s.a), "+v"(bf_[x].s.b));
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[x].s.a), "+v"(af[x].s.b));
#pragma unroll
    for (int x = 4; x < 8; ++x) { const uint32_t ad = aoff[x] + sb; W2V2_TR4(af[x], ad); }
    __builtin_amdgcn_sched_barrier(0);     // the second half of the dY fragments lands under the first 16 MFMAs
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        union { frag8_t v; uint32_t p[4]; } u;
        u.v = af[i].v;
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[i] = pair_sum_add<TE>(u.p[e], one2, bsum[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16<TE>(bf_[j].v, af[i].v, acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 4; x < 8; ++x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[x].s.a), "+v"(af[x].s.b));
    if (do_bias) {
#pragma unroll
      for (int i = 4; i < 8; ++i) {
        union { frag8_t v; uint32_t p[4]; } u;
        u.v = af[i].v;
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[i] = pair_sum_add<TE>(u.p[e], one2, bsum[i]);
      }
    }
#pragma unroll
    for (int i = 4; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16<TE>(bf_[j].v, af[i].v, acc[i][j]);
  }
#undef W2V2_TR4

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float s = bsum[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int m = m0 + wm * 128 + i * 16 + (lane & 15);
      if ((lane >> 4) == 0 && m < P.n_out) P.dbias[m] = s;
    }
  }
  // coalesced f32 tile store through LDS: four passes of 64 rows (fragments (p & 1) * 4 .. +3 of the waves wm == p >> 1)
  float* stagef = reinterpret_cast<float*>(smem_raw);
  constexpr int PITCH = BN + 4;
  const int frow = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {             // fully unrolled: the accumulator indices must stay static
    __syncthreads();
    if (wm == (pass >> 1)) {
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[(pass & 1) * 4 + i4][j];
          *reinterpret_cast<float4*>(stagef + (i4 * 16 + frow) * PITCH + wn * 64 + j * 16 + fk * 4) =
              make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 512 * it;                 // 64 rows x 64 float4 chunks
      const int r = c >> 6, ch = c & 63;
      const int m = m0 + pass * 64 + r, n = n0 + ch * 4;
      if (m < P.n_out && n + 4 <= P.n_in)
        store16_wt(P.dW + (int64_t)m * P.ld_dw + n, *reinterpret_cast<const uint4*>(stagef + r * PITCH + ch * 4));
    }
  }
}

template <typename TE>
static void wgrad_launch(const WgArgs& a, int tiles, bool ring, bool ring4, hipStream_t st) {
  if (ring4) {
    constexpr size_t lds = (size_t)4 * 32 * (256 + 256) * sizeof(bf16_t);   // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_ring4_kernel<TE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_grouped_ring4_kernel<TE>), dim3(tiles), dim3(512), lds, st, a);
  } else if (ring) {
    constexpr size_t lds = (size_t)3 * 64 * (256 + 128) * sizeof(bf16_t);   // 144 KiB
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_ring_kernel<TE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_grouped_ring_kernel<TE>, dim3(tiles), dim3(512), lds, st, a);
  } else {
    constexpr size_t lds = (size_t)2 * 2 * 64 * 128 * sizeof(bf16_t);   // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel<TE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_grouped_kernel<TE>, dim3(tiles), dim3(256), lds, st, a);
  }
}

static int g_wgrad_force = 0;
extern "C" int w2v2_tune_wgrad_kernel(int family) {
  const int old = g_wgrad_force;
  if (family >= 0 && family <= 6) g_wgrad_force = family;
  return old;
}

extern "C" int w2v2_wgrad_grouped(const w2v2_wgrad_problem* probs, int n, int tokens, int tokens_padded, int dtype,
                                  void* stream) {
  W2V2_REQUIRE(probs && n > 0 && n <= WG_MAXP, "wgrad_grouped: need 1..%d problems", WG_MAXP);
  W2V2_REQUIRE(tokens > 0 && tokens_padded >= tokens && tokens_padded % 64 == 0,
               "wgrad_grouped: tokens_padded must be tokens rounded up to a multiple of 64");
  WgArgs a;
  int tiles = 0;
  // 256-row ring kernel unless every problem is narrower than one 256-row tile (or W2V2_WGRAD_V1 is set)
  int max_out = 0;
  for (int i = 0; i < n; ++i) max_out = probs[i].n_out > max_out ? probs[i].n_out : max_out;
  static const bool env_ring = getenv("W2V2_WGRAD_V1") == nullptr, env_ring4 = getenv("W2V2_NO_WGRAD4") == nullptr;   // A/B switches
  static const bool env_phased = getenv("W2V2_NO_WGRAD_PH") == nullptr;
  const int force = g_wgrad_force;                   // tools / tests: 1 = 128x128, 2 = 256x128 ring, 3 = 256x256x32 ring, 4 = phased
  const bool ring = force ? force >= 2 : (max_out > 128 && env_ring);
  // 256x256 tiles when they alone fill >= 80 % of the CUs (e.g. the 8 problems of two w2v2-base blocks: 216 tiles)
  int64_t t4 = 0;
  for (int i = 0; i < n; ++i) t4 += cdiv(probs[i].n_out, 256) * cdiv(probs[i].n_in, 256);
  static int ncu_cached = 0;                         // (one device per process: queried once, not on every launch)
  if (ncu_cached == 0) {
    int dev = 0, v = 0;
    (void)hipGetDevice(&dev);
    ncu_cached = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int ncu = ncu_cached;
  // ... and when their rounds over the chip cost less than the rounds of 256x128 tiles (a 256x256 tile takes ~1.7x
  // the time of a 256x128 one; problems narrower than 256 rows -- the 128-channel Res2Net TDNNs of ECAPA -- leave
  // half of a 256x256 tile empty, so a mixed group can need MORE time on the large tiles)
  int64_t t3 = 0;
  for (int i = 0; i < n; ++i) t3 += cdiv(probs[i].n_out, 256) * cdiv(probs[i].n_in, 128);
  const bool ring4 = force ? force >= 3
                           : (ring && t4 * 10 >= (int64_t)ncu * 8 && cdiv(t4, ncu) * 17 <= cdiv(t3, ncu) * 10 && env_ring4);
  // the same 256x256 tiles on the phased kernel (wgrad_phased.hip): bit-equal results, anti-phase wave groups
  const bool phased = force ? force >= 4 : (ring4 && env_phased);
  const int late = force >= 4 ? force - 4 : 0;
  const int bm = ring ? 256 : 128;
  const int bn = ring4 ? 256 : 128;
  for (int i = 0; i < n; ++i) {
    const w2v2_wgrad_problem& q = probs[i];
    W2V2_REQUIRE(q.dY && q.X && q.dW, "wgrad_grouped: null operand in problem %d", i);
    W2V2_REQUIRE(q.n_out >= 8 && q.n_in >= 8 && q.n_out % 8 == 0 && q.n_in % 4 == 0 && q.n_in % 8 == 0,
                 "wgrad_grouped: n_out/n_in must be multiples of 8 (problem %d)", i);
    W2V2_REQUIRE(q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 && q.ld_dw % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(q.dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.X) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(q.dW) & 15) == 0,
                 "wgrad_grouped: operands must be 16-byte aligned (problem %d)", i);
    WgProblem& P = a.p[i];
    P.dY = (const bf16_t*)q.dY; P.X = (const bf16_t*)q.X; P.dW = q.dW; P.dbias = q.dbias;
    P.ld_dy = q.ld_dy; P.ld_x = q.ld_x; P.ld_dw = q.ld_dw;
    P.n_out = q.n_out; P.n_in = q.n_in;
    P.tiles_n = (int)cdiv(q.n_in, bn);
    P.tile_begin = tiles;
    tiles += (int)cdiv(q.n_out, bm) * P.tiles_n;
  }
  for (int i = n; i < WG_MAXP; ++i) a.p[i] = a.p[0];
  a.n_problems = n;
  a.total_tiles = tiles;
  a.ktiles = tokens_padded / 64;
  W2V2_REQUIRE(dtype == W2V2_BF16 || dtype == W2V2_F16, "wgrad_grouped: needs a 16-bit activation dtype (got %d)", dtype);
  if (phased) w2v2_launch_wgrad_phased(a, dtype, tiles, late, as_stream(stream));
  else W2V2_DISPATCH_16(dtype, "wgrad_grouped", wgrad_launch<AT>(a, tiles, ring, ring4, as_stream(stream)););
  W2V2_CHECK_LAUNCH("wgrad_grouped");
  return 0;
}
