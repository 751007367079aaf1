// gemm_f32_dma.hip -- exact-f32 GEMM on v_mfma_f32_32x32x2_f32 with the operands staged by LDS-DMA (round 6).
//
// gemm_f32.hip stages both operands through registers (8 x 16-byte loads per thread and K tile, then 8-32 LDS stores
// behind the MFMAs of the tile) with a one-tile prefetch distance, on 128x128 / 64x64 tiles only: on ECAPA-TDNN's
// products (BASELINE configs[4]) that left 19800 x 1024 x 1024 at 101 TFLOP/s (2.42 rounds of 128-row tiles, so the
// 64x64 tile with two LDS reads per MFMA ran instead) and the 128-channel Res2Net products at 47-68 TFLOP/s (a
// 4-12 trip K loop bound by the latency of its one-deep prefetch).  This kernel:
//   * block tile (32 FI) x 128 x 32, FI = 1..5: the four waves stand side by side along N (32 columns each), every wave
//     owns all 32 FI rows -- FI is picked per product so that the grid fills its last round (160-row tiles: 19800 x 1024
//     = 992 tiles = 1.94 rounds of 512 slots; 19800 x 3072 = 5.81 rounds instead of 7.27 rounds of 128-row tiles);
//   * operand images written by `global_load_lds_dwordx4` (1 KiB pieces, no staging registers, no LDS stores), in an
//     NST-stage ring with a COUNTED vmcnt wait (NST = 3 for the small tiles: two K tiles in flight while one is
//     multiplied; NST = 2 for FI >= 3 where two workgroups per CU leave 80 KiB each);
//   * K-contiguous operands ([rows][K]) land as [row][8 chunks of 4 k], the chunk index XOR-swizzled by (row >> 1) & 7
//     on the SOURCE side; a lane reads ONE ds_read_b128 per 32-row block and four k-steps -- its 4 values feed 4
//     consecutive MFMAs, so the MFMA k-pairs are (8q + e, 8q + 4 + e) instead of (2s, 2s + 1): still a plain fmaf
//     chain over all of K per output, in a fixed order that is not the ascending one;
//   * K-major operands ([K][rows]) land as [k][rows]; a lane reads 4 (2, 1) consecutive rows of one k with one
//     ds_read_b128 (b64, b32): MFMA block e of the group then holds rows 4 rl + e -- the row permutation is undone in
//     the epilogue's row index, nothing moves;
//   * every edge (rows beyond M / N, the K tail of a tile or of a split) is a lane whose DMA source is a 16-byte zero
//     constant: no scalar fallback path in the loop.
// Host side: w2v2_launch_gemm_f32 (gemm_f32.hip) sends a product here when both operands are plain 16-byte aligned
// matrices with K % 4 == 0 (and M % 4 / N % 4 for K-major ones); anything else stays on the register-staged kernel.
#include "gemm_common.h"
#include <utility>

typedef __attribute__((ext_vector_type(16))) float f32x16;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loop bodies that need their index at compile time
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __attribute__((aligned(16))) float g_f32_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// One 1 KiB LDS-DMA piece: 16 bytes per lane from `src` to LDS byte address lds + 16 lane.  Written as inline assembly on
// purpose: for the builtin (__builtin_amdgcn_global_load_lds) the compiler's waitcnt pass treats every later LDS read as
// possibly aliasing the piece and puts `s_waitcnt vmcnt(0)` in front of the first ds_read behind it -- in this loop that
// is the fragment read of the NEXT k-group, so every K tile waited out the full latency of the pieces it had just
// issued (tile kt + 1's, which nothing reads before the next barrier).  The ring's own counted waits below order the
// pieces against their readers.  Nothing else in this kernel uses M0.
__device__ __forceinline__ void f32_dma16(const float* src, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(src) : "memory");
}

template <int N> __device__ __forceinline__ void f32_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool TA, bool TB, int FI, int NST>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmArgs g) {
  constexpr int BM = 32 * FI, BN = 128, BK = 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PW = FI + 4;                        // 1 KiB DMA pieces per wave and stage: FI of A, 4 of B
  constexpr int SPC = 2;                            // MFMAs between two pieces (4 and 16 FI / PW measured the same +-1 %)
  static_assert(SPC * PW <= 16 * FI, "the pieces of a tile fit between its MFMAs");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kl = lane >> 5, rl = lane & 31;
  const int ntile = g.tiles_m * g.tiles_n;
  const int tile = (g.xcd_tiles & 1) ? xcd_remap(blockIdx.x, ntile) : (int)blockIdx.x;
  const bool dbg_nodma = g.xcd_tiles & 2, dbg_nobar = g.xcd_tiles & 4, dbg_nowait = g.xcd_tiles & 8;      // tools only (garbage results)
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int z0 = z / g.batch_inner, z1 = z - z0 * g.batch_inner;
  const int split = blockIdx.y;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;
  const float* Ab = reinterpret_cast<const float*>(g.A.ptr) + z0 * g.a_s0 + z1 * g.a_s1;
  const float* Bb = reinterpret_cast<const float*>(g.B.ptr) + z0 * g.b_s0 + z1 * g.b_s1;
  const float* zero = g_f32_zero16;

  // ---- DMA sources.  Piece p of this wave (p < FI: A, else B): pp[p] = this lane's 16 bytes of the NEXT K tile to be
  // issued (or the zero constant for a row beyond M / N), inc[p] = its advance per K tile in bytes (0 for a zero lane),
  // kof[p] = the lane's k offset inside a K tile (only the tail tile of a K range looks at it).  A steady-state piece is
  // then {s_mov m0, global_load_lds, 64-bit add}: the first version formed every source from (kt, validity, tail) with
  // ~25 mostly scalar instructions and three branches per piece, between MFMAs that wait for them.
  const char* pp[PW];
  uint32_t inc[PW];
  int kof[PW];
  static_for<PW>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    const float* src = nullptr;
    if constexpr (p < FI) {
      const int piece = wave * FI + p;
      if constexpr (!TA) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        kof[p] = 4 * c;
        if (m0 + row < g.M) src = Ab + (int64_t)(m0 + row) * g.A.ld + kbeg + 4 * c;
      } else {
        const int L = piece * 64 + lane, krow = L / (BM / 4), cir = L - krow * (BM / 4);
        kof[p] = krow;
        if (m0 + 4 * cir < g.M) src = Ab + (int64_t)(kbeg + krow) * g.A.ld + m0 + 4 * cir;
      }
      inc[p] = src ? (uint32_t)(TA ? (int64_t)BK * g.A.ld * 4 : BK * 4) : 0u;
    } else {
      const int piece = wave * 4 + (p - FI);
      if constexpr (!TB) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        kof[p] = 4 * c;
        if (n0 + row < g.N) src = Bb + (int64_t)(n0 + row) * g.B.ld + kbeg + 4 * c;
      } else {
        const int krow = 2 * piece + kl;
        kof[p] = krow;
        if (n0 + 4 * rl < g.N) src = Bb + (int64_t)(kbeg + krow) * g.B.ld + n0 + 4 * rl;
      }
      inc[p] = src ? (uint32_t)(TB ? (int64_t)BK * g.B.ld * 4 : BK * 4) : 0u;
    }
    pp[p] = reinterpret_cast<const char*>(src ? src : zero);
  });
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // krem = valid k of the tile being issued (< BK only in the tail tile of the K range: those lanes read zeros)
  auto issue_piece = [&](int krem, uint32_t stage_lds, auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int off = p < FI ? p * 1024 : A_BYTES + (p - FI) * 1024;      // + the wave's part, in stage_lds
    const char* s = kof[p] < krem ? pp[p] : reinterpret_cast<const char*>(zero);
    f32_dma16(reinterpret_cast<const float*>(s), stage_lds + off);
    pp[p] += inc[p];
  };
  // LDS byte address of this wave's first A piece of stage st (its B pieces: + A_BYTES - the A part + the B part)
  auto stage_addr = [&](int st) -> uint32_t { return __builtin_amdgcn_readfirstlane(lds0 + st * STAGE); };
  const int wa_off = wave * FI * 1024, wb_off = wave * 4 * 1024;

  // ---- fragment addresses.  K-contiguous image: row * 128 + ((2 q + kl) ^ s) * 16, s = (row >> 1) & 7 = (rl >> 1) & 7 for
  // every 32-row block; 2 q + kl = 2 q ^ kl, so the lane part is ((kl ^ s) << 4) and q enters as XOR (q << 5)
  int offq[4];
  {
    const int x = (kl ^ ((rl >> 1) & 7)) << 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) offq[q] = rl * 128 + (x ^ (q << 5));
  }
  // K-major image: k = 8 q + 4 kl + e; lane part = the 4 kl rows + its rl-th group of consecutive rows
  const int akm = 4 * kl * (BM * 4), bkm = A_BYTES + 4 * kl * 512 + (32 * wave + rl) * 4;

  f32x16 acc[FI];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  float fa[2][FI][4], fb[2][4];
  auto frags = [&](int set, const char* st, int q) {
    if constexpr (!TA) {
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(st + offq[q] + i * 4096);
        fa[set][i][0] = v.x; fa[set][i][1] = v.y; fa[set][i][2] = v.z; fa[set][i][3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const char* rowp = st + akm + (8 * q + e) * (BM * 4);
        if constexpr (FI >= 4) {
          const float4 v = *reinterpret_cast<const float4*>(rowp + rl * 16);
          fa[set][0][e] = v.x; fa[set][1][e] = v.y; fa[set][2][e] = v.z; fa[set][3][e] = v.w;
          if constexpr (FI == 5) fa[set][4][e] = *reinterpret_cast<const float*>(rowp + 512 + rl * 4);
        } else if constexpr (FI >= 2) {
          const float2 v = *reinterpret_cast<const float2*>(rowp + rl * 8);
          fa[set][0][e] = v.x; fa[set][1][e] = v.y;
          if constexpr (FI == 3) fa[set][2][e] = *reinterpret_cast<const float*>(rowp + 256 + rl * 4);
        } else {
          fa[set][0][e] = *reinterpret_cast<const float*>(rowp + rl * 4);
        }
      }
    }
    if constexpr (!TB) {
      const float4 v = *reinterpret_cast<const float4*>(st + A_BYTES + wave * 4096 + offq[q]);
      fb[set][0] = v.x; fb[set][1] = v.y; fb[set][2] = v.z; fb[set][3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) fb[set][e] = *reinterpret_cast<const float*>(st + bkm + (8 * q + e) * 512);
    }
  };

  // ---- the ring: tiles kt + 1 .. kt + NST - 1 in flight while tile kt is multiplied.
  // The pieces of tile kt + NST - 1 are issued BETWEEN the MFMAs of tile kt, one behind every second MFMA from the start
  // of the tile: a piece costs 60-180 issue cycles (MI355X_MICROARCH.md, LDS-DMA) and nine of them in one block behind the
  // barrier kept the wave -- and, with the two resident workgroups in step, the matrix pipe -- idle for that long in
  // every K tile; a v_mfma_f32_32x32x2_f32 holds the pipe for 64 cycles, which covers a piece per two MFMAs.
  auto piece_lds = [&](uint32_t sa, auto pc) -> uint32_t {
    constexpr int p = decltype(pc)::value;
    return sa + (p < FI ? wa_off : wb_off);
  };
  static_for<NST - 1>([&](auto sc) {
    constexpr int s_ = decltype(sc)::value;
    if (s_ < nk) {
      const uint32_t sa = stage_addr(s_);
      const int krem = kend - (kbeg + s_ * BK);
      static_for<PW>([&](auto pc) { issue_piece(krem, piece_lds(sa, pc), pc); });
    }
  });
  int st = 0;                                        // stage of tile kt
  for (int kt = 0; kt < nk; ++kt) {
    if (dbg_nowait) {
    } else if (NST > 2 && kt + NST - 1 <= nk) f32_wait_vmcnt<(NST - 2) * PW>();   // NST - 1 tiles issued, the oldest must be in
    else f32_wait_vmcnt<0>();
    if (!dbg_nobar) __builtin_amdgcn_s_barrier();    // tile kt visible to all; stage (kt - 1) % NST free again
    const int nt = kt + NST - 1;                     // the tile whose pieces go out under this one, into the stage just freed
    const bool more = nt < nk && !dbg_nodma;
    const uint32_t sa = stage_addr(st == 0 ? NST - 1 : st - 1);
    const int krem = kend - (kbeg + nt * BK);
    const char* sb = smem + st * STAGE;
    frags(0, sb, 0);
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q < 3) frags((q + 1) & 1, sb, q + 1);
      __builtin_amdgcn_sched_barrier(0);
      static_for<4 * FI>([&](auto jc) {
        constexpr int j = decltype(jc)::value, e = j / FI, i = j % FI;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[q & 1][e], fa[q & 1][i][e], acc[i], 0, 0, 0);   // D[n][m]
        if constexpr ((q * 4 * FI + j) % SPC == SPC - 1) {
          const int idx = (q * 4 * FI + j) / SPC;    // a piece behind every SPC-th MFMA while there are pieces
          if (more) {
            __builtin_amdgcn_sched_barrier(0);
            static_for<PW>([&](auto pc) {
              if (idx == decltype(pc)::value) issue_piece(krem, piece_lds(sa, pc), pc);
            });
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    st = (st + 1 == NST) ? 0 : st + 1;
  }

  // ---- epilogue.  D[n][m]: lane (kl, rl), register r: n = 8 (r >> 2) + 4 kl + (r & 3); m = the lane's row of block i
  auto mrow = [&](int i) -> int {
    if constexpr (!TA) return 32 * i + rl;
    else if constexpr (FI >= 4) return i < 4 ? 4 * rl + i : 128 + rl;
    else if constexpr (FI >= 2) return i < 2 ? 2 * rl + i : 64 + rl;
    else return rl;
  };
  float* Cz = reinterpret_cast<float*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const float* auxz = g.aux ? reinterpret_cast<const float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  float* auxo = g.aux ? reinterpret_cast<float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  const int nw = n0 + 32 * wave;
  if (g.atomic && (g.epilogue == W2V2_EPI_NONE || g.epilogue == W2V2_EPI_BIAS)) {
    // split-K / accumulating products: through LDS so that one atomic instruction runs ALONG rows (see gemm_f32.hip)
    constexpr int SP = 33;
    static_assert(4 * 32 * SP * 4 <= 2 * STAGE, "staging fits the ring");
    __syncthreads();                                 // every wave is done with the operand images
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * SP);
    const bool add_bias = g.epilogue == W2V2_EPI_BIAS && split == 0 && bias != nullptr;
    const int n = nw + rl;
    const float bv = (add_bias && n < g.N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[rl * SP + 8 * (r >> 2) + 4 * kl + (r & 3)] = acc[i][r] * g.alpha;
      __syncthreads();
#pragma unroll 8
      for (int r2 = 0; r2 < 16; ++r2) {
        const int rr = 2 * r2 + kl;                  // stage row rr = lane rr's row of block i
        int mm;
        if constexpr (!TA) mm = 32 * i + rr;
        else if constexpr (FI >= 4) mm = i < 4 ? 4 * rr + i : 128 + rr;
        else if constexpr (FI >= 2) mm = i < 2 ? 2 * rr + i : 64 + rr;
        else mm = rr;
        if (m0 + mm < g.M && n < g.N) unsafeAtomicAdd(Cz + (int64_t)(m0 + mm) * g.ldc + n, stage[rr * SP + rl] + bv);
      }
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v4[4] = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
      epilogue_store4<float>(g, Cz, auxz, auxo, bias, m0 + mrow(i), nw + q * 8 + kl * 4, v4, split == 0);
    }
}

template <bool TA, bool TB, int FI, int NST>
static void launch_dma(GemmArgs a, int M, int N, int split, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)NST * (32 * FI + 128) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_dma_kernel<TA, TB, FI, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
    if (getenv("W2V2_F32_OCC")) {                     // tools: resident workgroups per CU of this instantiation
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_f32_dma_kernel<TA, TB, FI, NST>, 256, lds);
      fprintf(stderr, "gemm_f32_dma<%d,%d,FI=%d,NST=%d>: %zu B LDS, %d workgroups per CU\n", (int)TA, (int)TB, FI, NST, lds, nb);
    }
  }
  a.tiles_m = (int)cdiv(M, 32 * FI);
  a.tiles_n = (int)cdiv(N, 128);
  dim3 grid(a.tiles_m * a.tiles_n, split, batch);
  hipLaunchKernelGGL((gemm_f32_dma_kernel<TA, TB, FI, NST>), grid, dim3(256), lds, st, a);
}

template <int FI, int NST>
static void launch_dma_layout(const GemmArgs& a, int M, int N, int split, int batch, hipStream_t st) {
  if (!a.A.trans && !a.B.trans) launch_dma<false, false, FI, NST>(a, M, N, split, batch, st);
  else if (!a.A.trans && a.B.trans) launch_dma<false, true, FI, NST>(a, M, N, split, batch, st);
  else if (a.A.trans && !a.B.trans) launch_dma<true, false, FI, NST>(a, M, N, split, batch, st);
  else launch_dma<true, true, FI, NST>(a, M, N, split, batch, st);
}

// fi = 1..5 (rows = 32 fi).  nst: the kernel is written for any ring depth; only the two-stage ring is instantiated -- a
// third stage (fi <= 2: 60 / 72 KiB) halves the resident workgroups and measured 5-20 % slower on every product.
void w2v2_launch_gemm_f32_dma(const GemmArgs& a, int M, int N, int split, int batch, int fi, int nst, hipStream_t st) {
#define F32_DMA_CASE(FI_, NST_) launch_dma_layout<FI_, NST_>(a, M, N, split, batch, st)
  switch (fi * 10 + nst) {
    case 12: F32_DMA_CASE(1, 2); break;
    case 22: F32_DMA_CASE(2, 2); break;
    case 32: F32_DMA_CASE(3, 2); break;
    case 42: F32_DMA_CASE(4, 2); break;
    default: F32_DMA_CASE(5, 2); break;
  }
#undef F32_DMA_CASE
}
