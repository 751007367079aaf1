// gemm_duo.hip -- 256x128x32 LDS-DMA ring GEMM for TWO independent four-wave workgroups per CU (see gemm_common.h).
#include "gemm_common.h"

// ------------------------------------------------------------------------------ 256 x 128 x 32, two workgroups per CU
// Why another family (round 5): the K = 768 products with heavy epilogues (FFN1: bias + GELU + GELU', two 60 MB output
// planes; dH: a 60 MB aux read and a 60 MB store) spend half their time in the epilogue -- ~30 VALU slots per element and
// a store drain at the HBM write rate (256 KiB per 256x256 tile at ~9 B/clk per CU = 12 us) -- with the matrix pipe idle:
// the 8-wave kernels own a CU alone (128-144 KiB of LDS), so nothing else can run under a tile's epilogue, and a second
// accumulator set does not fit (128 of a wave's 256 registers are accumulators).  Here a workgroup is FOUR waves (one per
// SIMD, 2 (m) x 2 (n), wave tile 128 x 64 as in the phased kernel: 12 fragment reads per 32 MFMAs) with a three-stage
// ring of 24 KiB K tiles (BK = 32: 72 KiB), so TWO workgroups share a CU, each SIMD holds one wave of each, and they run
// out of step: while one is in its epilogue (VALU, stores draining) the other's main loop has the matrix pipe, the LDS and
// the L2 feed to itself.  The workgroups of the second half of the grid start `stagger` later so that the two residents
// of a CU do not meet in the same phase.
//   * LDS rows are 64 bytes (32 k): chunk c of row r is stored at chunk c ^ ((-(r >> 2)) & 3) -- the b128 fragment read
//     (lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32: MI355X_MICROARCH.md, LDS) then touches 16 distinct 16-byte
//     slots per group; the B image is read with rows (rho >> 2) * 16 + j * 4 + (rho & 3) (register epilogue: a lane owns
//     16 consecutive columns), its map takes (r >> 4) instead of (r >> 2)
//   * a DMA piece is 16 rows x 64 bytes = 1 KiB; per stage a wave issues 4 A pieces + 2 B pieces (same counts as the
//     eight-wave ring kernel: `s_waitcnt vmcnt(6)` keeps two stages in flight)
__device__ __forceinline__ int swz32(int q) { return (4 - (q & 3)) & 3; }

template <typename TE, typename TC, int DBG = 0>     // DBG (tools): 1 = no epilogue
__global__ __launch_bounds__(256, 2) void gemm16_duo_256x128_kernel(const GemmArgs g, int stagger_ticks, int first_late) {
  constexpr int BM = 256, BN = 128, BK = 32, FM = 8, FN = 4;
  constexpr int STAGE = (BM + BN) * BK;           // elements per stage (A then B): 24 KiB
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int nk = g.K >> 5;
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const bf16_t* Bb = reinterpret_cast<const bf16_t*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;

  if ((int)blockIdx.x >= first_late && stagger_ticks > 0) {     // the second resident of every CU starts out of step
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(8);
  }

  const int frow = lane & 15, fk = lane >> 4;
  const int sw = swz32(frow >> 2);
  // fragment offsets (elements) inside a stage: A fragment i at + i * 16 * 32, B fragment j at + j * 4 * 32
  const int la = (wm * 128 + frow) * BK + ((fk ^ sw) << 3);
  const int lb = BM * BK + (wn * 64 + (frow >> 2) * 16 + (frow & 3)) * BK + ((fk ^ sw) << 3);

#pragma unroll 1
  for (int t0 = 0; t0 < ntile; t0 += G) {
    const int nchunk = min(G, ntile - t0);
    if ((int)blockIdx.x >= nchunk) break;
    const int tile = t0 + xcd_remap(blockIdx.x, nchunk);
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // Only 128 of the wave's 256 registers are VGPRs here (the accumulators take the AGPR half), so nothing the prologue or
    // the epilogue derives from the lane id may be hoisted across the main loop: the lane id is made opaque per tile.
    int lane_p = lane;
    asm volatile("" : "+v"(lane_p));
    const int c4 = lane_p & 3, r16 = lane_p >> 2;
    const bf16_t* ap[4];
    const bf16_t* bp[2];
    {
      int ca[4], cb[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) ca[j] = (c4 ^ swz32(r16 >> 2)) << 3;          // A rows: (row >> 2) & 3 = (r16 >> 2) & 3
#pragma unroll
      for (int j = 0; j < 2; ++j) cb[j] = (c4 ^ swz32(wave * 2 + j)) << 3;      // B rows: (row >> 4) & 3 = piece & 3
      tile_ptrs<4>(g.A, Ab, m0, BM, g.M, wave * 64 + r16, 16, ca, ap);
      tile_ptrs<2>(g.B, Bb, n0, BN, g.N, wave * 32 + r16, 16, cb, bp);
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // piece p of the 6 DMA pieces of this wave for one stage: A0..A3, B0, B1 (1 KiB each)
    auto stage_piece = [&](bf16_t* base, int kt, int p) {
      if (p < 4)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(ap[p] + kt * BK), (lvoid_t*)(base + (wave * 4 + p) * 16 * BK), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gvoid_t*)(bp[p - 4] + kt * BK),
                                         (lvoid_t*)(base + BM * BK + (wave * 2 + p - 4) * 16 * BK), 16, 0, 0);
    };
    auto stage = [&](bf16_t* base, int kt) {
#pragma unroll
      for (int p = 0; p < 6; ++p) stage_piece(base, kt, p);
    };
    // ONE loop body (stage offsets rotate in scalar registers) and accumulators pinned in AGPRs, updated in place by
    // `mfma16_agpr`: with three unrolled ring steps, or a branch that selects between a loading and a non-loading body, the
    // register allocator gives the paths different accumulator assignments and joins them with ~128 copies (and spills).
    // A stage is consumed by the twelve fragment reads at the head of its step (BK = 32 is one MFMA k-step), so its
    // buffer is FREE once every wave's reads have returned: the DMA pieces of K tile kt + 3 go into the buffer of K tile kt
    // behind a second barrier, under the step's own MFMAs -- three K tiles (72 KiB, the workgroup's whole LDS) are in
    // flight while the matrix pipe works instead of two (a workgroup's feed rate is bytes in flight / L2 latency).
    __builtin_amdgcn_s_barrier();          // every wave has finished reading the previous tile's stages
    stage(smem, 0);
    if (nk > 1) stage(smem + STAGE, 1);
    if (nk > 2) stage(smem + 2 * STAGE, 2);
    int cur = 0;                            // element offset of stage kt % 3
    int kt = 0;                             // (nk >= 2: the host sends K >= 64 here; a do-while keeps ONE path into the epilogue)
#pragma unroll 1
    do {
      // stage kt has landed when at most the pieces of K tiles kt + 1, kt + 2 (6 each) are outstanding
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 3 < nk;
      bf16_t* base = smem + cur;
      frag8_t bfr[FN], af[FM];
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const frag8_t*>(base + lb + j * 4 * BK);
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const frag8_t*>(base + la + i * 16 * BK);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();         // every wave holds its fragments: the buffer may be refilled
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) mfma16_agpr<TE>(bfr[j], af[i], acc[i][j]);
        if (i < 6) {                        // the six DMA pieces of K tile kt + 3, one behind each MFMA group
          __builtin_amdgcn_sched_barrier(0);
          if (more) stage_piece(base, kt + 3, i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cur = cur + STAGE == 3 * STAGE ? 0 : cur + STAGE;
    } while (++kt < nk);

    TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
    TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
    const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
    int nc = n0 + wn * 64 + fk * 16, mr = m0 + wm * 128 + frow;
    int lane_e = lane;
    asm volatile("" : "+v"(nc), "+v"(mr), "+v"(lane_e));
    if constexpr ((DBG & 1) != 0) {           // tools: no epilogue (one store keeps the accumulators live)
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sum == 12345.678f) reinterpret_cast<TC*>(g.C)[0] = from_f32<TC>(sum);
      continue;
    }
    float cv0[8], cv1[8];
    load_col8(g, bias, nc, cv0);
    load_col8(g, bias, nc + 8, cv1);
    if constexpr (sizeof(TC) == 2) {       // (the host sends 16-bit outputs here only where lines_ok holds)
      W2V2_EPI_DISPATCH((epilogue_lines<TC, EPI, FM>(g, Cz, auxz, acc, mr - (lane_e & 15), nc, lane_e, cv0, cv1, nullptr)));
    } else {
      W2V2_EPI_DISPATCH((epilogue_direct<TC, EPI, FM, 2>(g, Cz, auxz, acc, mr, nc, cv0, cv1)));
    }
  }   // tile loop
}

template <typename TE, typename TC, int DBG = 0>
static void launch_duo(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * (256 + 128) * 32 * sizeof(bf16_t);   // 72 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_duo_256x128_kernel<TE, TC, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 128);
  const int tiles = a.tiles_m * a.tiles_n;
  const int ncu = w2v2_gemm_device_cus();
  static const char* stag_env = getenv("W2V2_DUO_STAGGER_US");          // tools: A/B of the start offset
  // half of a tile's period: the main loop at ~0.45 us per K tile of 32 when it has the CU alone
  const double stag_us = stag_env ? atof(stag_env) : 0.5 * (0.45 * (a.K / 32) + 4.0);
  dim3 grid(tiles < 2 * ncu ? tiles : 2 * ncu, 1, batch);
  W2V2_LAUNCH_MAYBE_TIMED((gemm16_duo_256x128_kernel<TE, TC, DBG>), grid, dim3(256), lds, st, a, (int)(stag_us * 100.0), ncu);
}

void w2v2_launch_duo_256x128(const GemmArgs& a, int dtype_ab, int dtype_c, int M, int N, int batch, hipStream_t st) {
  if (dtype_ab == W2V2_BF16) {
    if (dtype_c == W2V2_F32) launch_duo<bf16_t, float>(a, M, N, batch, st);
    else launch_duo<bf16_t, bf16_t>(a, M, N, batch, st);
  } else {
    static const char* dbg_env = getenv("W2V2_DUO_DBG");                // tools: 1 = no epilogue (fp16 in / fp16 out)
    if (dtype_c == W2V2_F32) launch_duo<f16_t, float>(a, M, N, batch, st);
    else if (dbg_env && atoi(dbg_env) == 1) launch_duo<f16_t, f16_t, 1>(a, M, N, batch, st);
    else launch_duo<f16_t, f16_t>(a, M, N, batch, st);
  }
}
