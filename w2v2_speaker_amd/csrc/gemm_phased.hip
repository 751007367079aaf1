// gemm_phased.hip -- 256x256x64 phased GEMM kernel (see gemm_common.h for the family map).
#include "gemm_common.h"

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
// wait until all but the n most recently issued DMA pieces of this wave have landed
__device__ __forceinline__ void wait_pieces(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}
// wait until all but the `newer` most recently issued quarters (2 DMA pieces each) of this wave have landed
__device__ __forceinline__ void wait_quarters(int newer) {
  if (newer >= 4) wait_vm<8>();
  else if (newer == 3) wait_vm<6>();
  else if (newer == 2) wait_vm<4>();
  else if (newer == 1) wait_vm<2>();
  else wait_vm<0>();
}

// DBG (tools only, w2v2_tune_gemm_debug): time attribution / placement experiments on the SAME kernel body --
//   1 = no DMA in the steady-state loop, 2 = no fragment reads, 4 = no MFMAs, 8 = no epilogue,
//   16 / 32 = one / both of the two DMA pieces of a phase are issued BETWEEN its MFMAs instead of in its read segment,
//   (host side) 64 = plain write-back epilogue stores, 128 = write-through ones, whatever w2v2_gemm chose
template <typename TE, typename TC, int DBG = 0>
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
        w2v2_dma16(((isa ? Ab : Bb) + (soff[q][j] + kt * 64)), (base + piece_row(q, j) * 64));
    };
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K tile 0 entirely, QA0 / QB0 of K tile 1 (issue order = consumption order)
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    if (nk > 1) { issue(0, 1); issue(1, 1); }
    w2v2_vmcnt0_visible();                           // QA0(0), QB0(0) landed (this wave's pieces; all six quarters: common.h)
    __builtin_amdgcn_s_barrier();                    // ... everyone's
    if (wr == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind from here on

    frag8_t af[4][2], b0[2][2], b1[2][2];
    constexpr bool NO_DMA = (DBG & 1) != 0, NO_READ = (DBG & 2) != 0, NO_MFMA = (DBG & 4) != 0;
    constexpr int LATE = (DBG >> 4) & 3;             // DMA pieces of a phase (of 2) issued between its MFMAs instead of in its read segment
    if constexpr (NO_READ) {                           // (fragments defined once: the MFMAs keep real operands)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          b0[j][kk] = *reinterpret_cast<const frag8_t*>(smem + (kk ? lb1 : lb0) + j * 4 * 64);
          b1[j][kk] = *reinterpret_cast<const frag8_t*>(smem + (kk ? lb1 : lb0) + (2 + j) * 4 * 64);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) af[i][kk] = *reinterpret_cast<const frag8_t*>(smem + (kk ? la1 : la0) + i * 16 * 64);
    }
    // one DMA piece of quarter q (tools: placement experiment DBG & 16)
    auto issue1 = [&](int q, int kt, int j) {
      const bool isa = q == 0 || q == 3;
      bf16_t* base = smem + (kt & 1) * BUF + (isa ? 0 : BM * 64);
      w2v2_dma16(((isa ? Ab : Bb) + (soff[q][j] + kt * 64)), (base + piece_row(q, j) * 64));
    };
    // 16 MFMAs of a phase: acc rows io.., columns jo.., fragments bsel; with DMA_IN_MFMA the two pieces of quarter q of K
    // tile kq go out behind the 4th and the 12th MFMA
#define W2V2_PH_MFMA(io_, jo_, bfr_, doq_, q_, kq_)                                                       \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
          if constexpr (!NO_MFMA) {                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
              acc[io_ + i][jo_ + j] = mfma16<TE>(bfr_[j][kk], af[i][kk], acc[io_ + i][jo_ + j]);          \
          }                                                                                                \
          if constexpr (LATE == 2 && !NO_DMA) {                                                            \
            if (i == 1 && (doq_)) {                                                                        \
              __builtin_amdgcn_sched_barrier(0);                                                           \
              issue1(q_, kq_, kk);                                                                         \
              __builtin_amdgcn_sched_barrier(0);                                                           \
            }                                                                                              \
          }                                                                                                \
          if constexpr (LATE == 1 && !NO_DMA) {                                                            \
            if (kk == 0 && i == 3 && (doq_)) {                                                             \
              __builtin_amdgcn_sched_barrier(0);                                                           \
              issue1(q_, kq_, 1);                                                                          \
              __builtin_amdgcn_sched_barrier(0);                                                           \
            }                                                                                              \
          }                                                                                                \
        }                                                                                                  \
      }
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const bf16_t* bufp = smem + (kt & 1) * BUF;
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      // ---------------- phase 1: read B0 + A lo; issue QB1(kt+1); MFMA A lo x B0
      if constexpr (!NO_READ) {
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
      }
      if constexpr (!NO_DMA) { if (more1) { if constexpr (LATE == 0) issue(2, kt + 1); else if constexpr (LATE == 1) issue1(2, kt + 1, 0); } }
      if constexpr (!NO_DMA) wait_pieces(2 + (more1 ? 6 - LATE : 0));              // QB1(kt) for phase 2
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      W2V2_PH_MFMA(0, 0, b0, more1, 2, kt + 1)
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 2: read B1; issue QA1(kt+1); MFMA A lo x B1
      if constexpr (!NO_READ) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b1[j][0] = *reinterpret_cast<const frag8_t*>(bufp + lb0 + (2 + j) * 4 * 64);
          b1[j][1] = *reinterpret_cast<const frag8_t*>(bufp + lb1 + (2 + j) * 4 * 64);
        }
      }
      if constexpr (!NO_DMA) { if (more1) { if constexpr (LATE == 0) issue(3, kt + 1); else if constexpr (LATE == 1) issue1(3, kt + 1, 0); } }
      if constexpr (!NO_DMA) wait_pieces(more1 ? 8 - LATE : 0);                    // QA1(kt) for phase 3
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      W2V2_PH_MFMA(0, 2, b1, more1, 3, kt + 1)
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 3: read A hi; issue QA0(kt+2); MFMA A hi x B1
      if constexpr (!NO_READ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i][0] = *reinterpret_cast<const frag8_t*>(bufp + la0 + (4 + i) * 16 * 64);
          af[i][1] = *reinterpret_cast<const frag8_t*>(bufp + la1 + (4 + i) * 16 * 64);
        }
      }
      if constexpr (!NO_DMA) { if (more2) { if constexpr (LATE == 0) issue(0, kt + 2); else if constexpr (LATE == 1) issue1(0, kt + 2, 0); } }
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      W2V2_PH_MFMA(4, 2, b1, more2, 0, kt + 2)
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---------------- phase 4: (operands in registers); issue QB0(kt+2); MFMA A hi x B0
      if constexpr (!NO_DMA) { if (more2) { if constexpr (LATE == 0) issue(1, kt + 2); else if constexpr (LATE == 1) issue1(1, kt + 2, 0); } }
      if constexpr (!NO_DMA) { if (more1) wait_pieces(4 + (more2 ? 4 - LATE : 0)); }   // QA0(kt+1), QB0(kt+1) for the next K tile's phase 1
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      W2V2_PH_MFMA(4, 0, b0, more2, 1, kt + 2)
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    }
#undef W2V2_PH_MFMA
    if (wr == 0) __builtin_amdgcn_s_barrier();       // group 0 catches up: every read of this tile's buffers is done

    TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
    TC* auxz = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
    const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
    // the output row / column of this lane are made opaque HERE: left to itself the compiler forms the epilogue's
    // 64-bit row addresses before the main loop and, at the 256-register limit, spills them -- and a kernel that
    // touches scratch at all pays ~8 us per dispatch (tools/probes/scratch_probe.hip)
    int nc = n0 + wc * 64 + fk * 16, mr = m0 + wr * 128 + frow;
    asm volatile("" : "+v"(nc), "+v"(mr));
    if constexpr ((DBG & 8) != 0) { if (g.M > 0) continue; }   // (tools: no epilogue; the test keeps the accumulators live)
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

static int g_w2v2_dbg = 0;            // tools only (w2v2_tune_gemm_debug): DBG variant of the phased kernel, fp16 in / fp16 out
template <typename TE, typename TC, int DBG = 0>
static void launch_ph(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (256 + 256) * 64 * sizeof(bf16_t);   // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm16_phased_256x256_kernel<TE, TC, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, 256);
  a.tiles_n = (int)cdiv(N, 256);
  const int tiles = a.tiles_m * a.tiles_n;
  const int ncu = w2v2_gemm_device_cus();
  dim3 grid(tiles < ncu ? tiles : ncu, 1, batch);
  W2V2_LAUNCH_MAYBE_TIMED((gemm16_phased_256x256_kernel<TE, TC, DBG>), grid, dim3(512), lds, st, a);
}
template <typename TE, typename TC>
static void launch_ph_dbg(GemmArgs a, int M, int N, int batch, hipStream_t st) {
  if (g_w2v2_dbg & 192) a.wt_stores = (g_w2v2_dbg & 128) ? 1 : 0;    // tools: bit 6 forces plain, bit 7 write-through stores
  const int g_w2v2_dbg = ::g_w2v2_dbg & 63;
  if (g_w2v2_dbg == 0 && a.late_dma) return launch_ph<TE, TC, 32>(a, M, N, batch, st);   // the product's second variant
  if constexpr (std::is_same<TE, f16_t>::value && std::is_same<TC, f16_t>::value) {
    switch (g_w2v2_dbg) {
      case 1: return launch_ph<TE, TC, 1>(a, M, N, batch, st);
      case 2: return launch_ph<TE, TC, 2>(a, M, N, batch, st);
      case 4: return launch_ph<TE, TC, 4>(a, M, N, batch, st);
      case 8: return launch_ph<TE, TC, 8>(a, M, N, batch, st);
      case 9: return launch_ph<TE, TC, 9>(a, M, N, batch, st);
      case 11: return launch_ph<TE, TC, 11>(a, M, N, batch, st);
      case 13: return launch_ph<TE, TC, 13>(a, M, N, batch, st);
      case 16: return launch_ph<TE, TC, 16>(a, M, N, batch, st);
      case 24: return launch_ph<TE, TC, 24>(a, M, N, batch, st);
      case 48: return launch_ph<TE, TC, 0>(a, M, N, batch, st);    // tools: the read-segment placement whatever the host chose
      case 40: return launch_ph<TE, TC, 40>(a, M, N, batch, st);

      default: break;
    }
  }
  launch_ph<TE, TC>(a, M, N, batch, st);
}
void w2v2_launch_phased_256x256(const GemmArgs& a, int dtype_ab, int dtype_c, int M, int N, int batch, hipStream_t st) {
  if (dtype_ab == W2V2_BF16) {
    if (dtype_c == W2V2_F32) launch_ph_dbg<bf16_t, float>(a, M, N, batch, st);
    else launch_ph_dbg<bf16_t, bf16_t>(a, M, N, batch, st);
  } else {
    if (dtype_c == W2V2_F32) launch_ph_dbg<f16_t, float>(a, M, N, batch, st);
    else launch_ph_dbg<f16_t, f16_t>(a, M, N, batch, st);
  }
}

extern "C" int w2v2_tune_gemm_debug(int bits) {
  const int old = g_w2v2_dbg;
  g_w2v2_dbg = bits;
  return old;
}
