// gemm_ring.hip -- 256x128x64 three-stage LDS-DMA ring GEMM (see gemm_common.h for the family map).
#include "gemm_common.h"

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

template <typename TE, typename TC, bool DBG = false>
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
        if (mi < g.M) {
          if (g.wt_stores) store16_wt(Cdef + (int64_t)mi * g.ldc + pend_n, pend[q]);
          else *reinterpret_cast<uint4*>(Cdef + (int64_t)mi * g.ldc + pend_n) = pend[q];
        }
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
      w2v2_dma16((ap[j] + ka), (ad + j * 8 * 64));
#pragma unroll
    for (int j = 0; j < 2; ++j)
      w2v2_dma16((bp[j] + kb), (bd + j * 8 * 64));
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
      w2v2_dma16((ap[p] + koff_a(kt)), (base + (wave * 4 + p) * 8 * 64));
    else
      w2v2_dma16((bp[p - 4] + koff_b(kt)), (base + BM * 64 + (wave * 2 + p - 4) * 8 * 64));
  };
  // multiply stage `base`; when kload >= 0 the DMA pieces of K tile kload go to `nxt`, spread over the MFMA groups
  // (issued back to back behind the barrier they keep both waves of a SIMD in the queue-limited DMA issue)
  // tools only (w2v2_tune_gemm_ring_debug, carried in g.late_dma for ring launches; results are garbage):
  //   1 = no DMA pieces in the steady-state loop, 2 = no barrier in the loop, 4 = no vmcnt wait in the loop, 8 = no fragment reads
  const int dbg = DBG ? g.late_dma : 0;
  auto compute = [&](const bf16_t* base, bf16_t* nxt, int kload) {
    if (dbg & 1) kload = -1;
    const bf16_t* a0 = base + aoff + lo0;
    const bf16_t* a1 = base + aoff + lo1;
    const bf16_t* b0 = base + boff + lb0;
    const bf16_t* b1 = base + boff + lb1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag8_t af[FM], bfr[FN];
      if (DBG && (dbg & 8)) {                      // registers as they are (no load, nothing kept live for it)
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" : "=v"(af[i]));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(bfr[j]));
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const frag8_t*>((kk ? a1 : a0) + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const frag8_t*>((kk ? b1 : b0) + j * 4 * 64);
      }
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
    if (!(dbg & 4)) { if (kt + 1 < nk) wait_vmcnt<6>(); else wait_vmcnt<0>(); }   \
    if (!(dbg & 2)) __builtin_amdgcn_s_barrier();                  \
    compute(cur, nxt, kt + 2 < nk ? kt + 2 : -1);                  \
    ++kt;                                                          \
  }
  w2v2_vmcnt0_visible();                 // (common.h: keeps the compiler's own vmcnt(0) out of the loop; nothing of this wave's is in flight here but the previous tile's epilogue loads / non-deferred stores)
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
  if (DBG && (dbg & 32)) { if (g.M > 0) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3])); continue; } }   // (tools: no epilogue)
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

template <typename TE, typename TC, bool DBG>
static void launch_ring_v(GemmArgs a, int M, int N, int batch, bool persistent, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * (256 + 128) * 64 * sizeof(bf16_t);   // 144 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_ring_256x128_kernel<TE, TC, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 128);
  const int ncu = persistent ? w2v2_gemm_device_cus() : (1 << 30);
  const int tiles = a.tiles_m * a.tiles_n;
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  W2V2_LAUNCH_MAYBE_TIMED((gemm16_ring_256x128_kernel<TE, TC, DBG>), grid, dim3(512), lds, st, a);
}
template <typename TE, typename TC>
static void launch_ring(GemmArgs a, int M, int N, int batch, bool persistent, hipStream_t st) {
  if (a.late_dma != 0 && sizeof(TC) == 2 && sizeof(TE) == 2 && std::is_same<TE, f16_t>::value)
    launch_ring_v<f16_t, f16_t, true>(a, M, N, batch, persistent, st);      // tools: attribution variants, fp16 in / fp16 out only
  else launch_ring_v<TE, TC, false>(a, M, N, batch, persistent, st);
}

static int g_ring_dbg = 0;            // tools only: time-attribution variants of the ring kernel's K loop (garbage results)
extern "C" int w2v2_tune_gemm_ring_debug(int bits) {
  const int old = g_ring_dbg;
  g_ring_dbg = bits & 63;          // bit 4: the attribution kernel with nothing removed (its own baseline); bit 5: no epilogue
  return old;
}
void w2v2_launch_ring_256x128(const GemmArgs& a_in, int dtype_ab, int dtype_c, int M, int N, int batch, bool persistent,
                              hipStream_t st) {
  GemmArgs a = a_in;
  a.late_dma = g_ring_dbg;
  if (dtype_ab == W2V2_BF16) {
    if (dtype_c == W2V2_F32) launch_ring<bf16_t, float>(a, M, N, batch, persistent, st);
    else launch_ring<bf16_t, bf16_t>(a, M, N, batch, persistent, st);
  } else {
    if (dtype_c == W2V2_F32) launch_ring<f16_t, float>(a, M, N, batch, persistent, st);
    else launch_ring<f16_t, f16_t>(a, M, N, batch, persistent, st);
  }
}
