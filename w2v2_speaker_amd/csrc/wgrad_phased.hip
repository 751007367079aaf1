// wgrad_phased.hip -- grouped weight gradients on 256x256x64 tiles with the two wave groups of a workgroup in ANTI-PHASE
// (the structure of gemm16_phased_256x256_kernel, gemm_phased.hip, on K-major operands).
//
//   dW_p[o][i] = sum_t dY_p[t][o] * X_p[t][i]      dbias_p[o] = sum_t dY_p[t][o]
//
// Why: the 256x256x32 ring kernel (wgrad.hip) runs both waves of a SIMD through the same wait -> read -> MFMA sequence
// behind one barrier per 32 MFMAs; PMC round 3: matrix pipe 38 % busy, 1.64 ms per step in these launches.  Here a K tile
// of 64 tokens is four phases of 16 MFMAs; group 1 (waves 4-7) runs one barrier behind group 0 (waves 0-3), so on every
// SIMD one wave multiplies while its partner issues fragment reads, DMA pieces and waits.
//
// Layout.  Both operands are K-major ([token][feature]); they stay that way in LDS and the fragments come out of the
// hardware transpose read ds_read_b64_tr_b16 (16 lanes x 8 B = a 4 (k) x 16 (m) block, lane i receives column i).  A K-tile
// buffer (64 KiB, two of them) is four QUARTERS of 16 KiB = [64 tokens][128 features], pitch 256 B:
//     QA0 = dY columns m0 + [0, 128)   QA1 = m0 + [128, 256)   QB0 = X columns n0 + [0, 128)   QB1 = n0 + [128, 256)
// so that a DMA piece (1 KiB per wave instruction) is four token rows x 256 contiguous bytes of HBM.  The waves are mapped
// onto the tile so that every wave needs one half of each quarter pair per phase: wave (wr, wc), wr = wave >> 2 (group),
// wc = wave & 3, owns rows (n_out)  m = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 .. + 15, i = 0..7   (A fragment i)
//                 and columns (n_in) n = (j >> 1) * 128 + wc * 32 + (j & 1) * 16 .. + 15, j = 0..3  (B fragment j).
// The 32-byte segments of a 256-byte row are XOR-swizzled with f(r) = (r & 3) | ((r >> 1) & 4) (r = token row) on the DMA
// source side and again on the read: the eight rows a half-wave's transposing read touches land on eight different
// segments = all 64 banks once (the map of the ring kernels, wgrad.hip).
//
// Schedule per K tile (quarters refilled as soon as both groups have read them, `s_waitcnt vmcnt(8)` in steady state):
//   phase 1: read B0 (QB0) + A lo (QA0) | issue QB1(kt+1) | MFMA A lo x B0
//   phase 2: read B1 (QB1)              | issue QA1(kt+1) | MFMA A lo x B1
//   phase 3: read A hi (QA1)            | issue QA0(kt+2) | MFMA A hi x B1
//   phase 4: --                         | issue QB0(kt+2) | MFMA A hi x B0
// The k order of every accumulator is that of the ring kernels (token steps of 32 in ascending order), so the results
// are BIT-EQUAL to theirs (tests/test_kernels_gpu.py).  No split-K, no atomics: bitwise reproducible.
// Contract (w2v2_wgrad_grouped): token rows [tokens, tokens_padded) readable and zero, tokens_padded % 64 == 0; n_out, n_in
// multiples of 8; 16-byte aligned rows.
#include "wgrad_common.h"

template <int N> __device__ __forceinline__ void wgp_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void wgp_wait_pieces(int n) {     // all but the n most recently issued DMA pieces have landed
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
__device__ __forceinline__ void wgp_wait_quarters(int newer) {
  if (newer >= 4) wgp_wait_vm<8>();
  else if (newer == 3) wgp_wait_vm<6>();
  else if (newer == 2) wgp_wait_vm<4>();
  else if (newer == 1) wgp_wait_vm<2>();
  else wgp_wait_vm<0>();
}

union WgFrag { struct { short4v a, b; } s; frag8_t v; };
// one MFMA fragment = two transposing reads (token rows r and r + 4 of the lane's 8-row group); OFF = byte offset of the
// k-step / quarter.  asm: the builtin makes the compiler drain the DMA ring in front of every read (wgrad.hip).
template <int OFF> __device__ __forceinline__ void wgp_read(WgFrag& f, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.s.a) : "v"(addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.s.b) : "v"(addr), "n"(OFF + 1024) : "memory");
}
__device__ __forceinline__ void wgp_landed(WgFrag& f) {   // LDS returns in order: ties the registers to the wait
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.s.a), "+v"(f.s.b));
}

// LATE = how many of the two DMA pieces of a phase are issued between its MFMAs instead of in its read segment
template <typename TE, int LATE>
__global__ __launch_bounds__(512) void wgrad_grouped_phased_kernel(const WgArgs a) {
  constexpr int BM = 256, BN = 256;
  constexpr int QB = 64 * 128 * 2;                  // bytes per quarter
  constexpr int BUFB = 4 * QB;                      // bytes per K-tile buffer: QA0 | QA1 | QB0 | QB1
  constexpr int KSTEP = 32 * 256;                   // bytes between the two k-steps (32 token rows) of a quarter
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
  const int wr = wave >> 2, wc = wave & 3;
  const int nk = a.ktiles;                          // K tiles of 64 tokens

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

  // ---- DMA sources: quarter q in issue order {QA0, QB0, QB1, QA1}; piece j of this wave = token rows 8 wave + 4 j + rr
  const int rr = lane >> 4, pc = lane & 15;
  const int fsrc = rr | ((wave & 1) << 2);           // f(row): row & 3 = rr, row bit 3 = wave & 1 (rows 8 wave + 4 j + rr)
  const int lsrc = (pc >> 1) ^ fsrc;                 // logical 32-byte segment this lane fetches
  const int csrc = lsrc * 16 + (pc & 1) * 8;         // feature column inside the quarter
  int soff[4][2];                                    // element offsets from P.dY / P.X (+ kt * 64 * ld at issue)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool isa = q == 0 || q == 3;
    const int half = (q >= 2) ? 1 : 0;               // QB1, QA1 are the upper 128 features
    const int col = isa ? min(m0 + half * 128 + csrc, P.n_out - 8) : min(n0 + half * 128 + csrc, P.n_in - 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) soff[q][j] = (8 * wave + 4 * j + rr) * (int)(isa ? P.ld_dy : P.ld_x) + col;
  }
  // operand bases and K-tile strides live in SGPRs for the whole loop (made opaque: left alone, the compiler re-loads
  // them from the kernel-argument block in front of every DMA issue, and that s_load's `lgkmcnt(0)` also waits for the
  // fragment reads issued just before)
  const bf16_t* pa = P.dY;
  const bf16_t* pb = P.X;
  int64_t astep = 64 * P.ld_dy, bstep = 64 * P.ld_x;
  asm volatile("" : "+s"(pa), "+s"(pb), "+s"(astep), "+s"(bstep));
  auto issue = [&](int q, int kt) {                  // quarter q of K tile kt -> buffer kt & 1
    const bool isa = q == 0 || q == 3;
    const int region = q == 0 ? 0 : (q == 3 ? 1 : (q == 1 ? 2 : 3));
    char* dst = smem_raw + (kt & 1) * BUFB + region * QB + (8 * wave) * 256;
    const bf16_t* base = isa ? pa + kt * astep : pb + kt * bstep;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(base + soff[q][j]), (lvoid_t*)(dst + j * 4 * 256), 16, 0, 0);
  };
  auto issue1 = [&](int q, int kt, int j) {          // one piece of it
    const bool isa = q == 0 || q == 3;
    const int region = q == 0 ? 0 : (q == 3 ? 1 : (q == 1 ? 2 : 3));
    char* dst = smem_raw + (kt & 1) * BUFB + region * QB + (8 * wave) * 256;
    const bf16_t* base = isa ? pa + kt * astep : pb + kt * bstep;
    __builtin_amdgcn_global_load_lds((gvoid_t*)(base + soff[q][j]), (lvoid_t*)(dst + j * 4 * 256), 16, 0, 0);
  };
  // read-segment part of a phase's quarter / the part that goes between the MFMAs (behind MFMA number `after`)
  auto issue_early = [&](int q, int kt) {
    if constexpr (LATE == 0) issue(q, kt); else if constexpr (LATE == 1) issue1(q, kt, 0);
  };
#define W2V2_WGP_MFMA(io_, jo_, bfr_, doq_, q_, kq_)                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                     \
      _Pragma("unroll") for (int x = 0; x < 4; ++x) {                                                      \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                      \
          acc[io_ + x][jo_ + s] = mfma16<TE>(bfr_[s][kk].v, af[x][kk].v, acc[io_ + x][jo_ + s]);           \
        if constexpr (LATE == 2) {                                                                         \
          if (x == 1 && (doq_)) {                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            issue1(q_, kq_, kk);                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                             \
          }                                                                                                \
        }                                                                                                  \
        if constexpr (LATE == 1) {                                                                         \
          if (kk == 0 && x == 3 && (doq_)) {                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            issue1(q_, kq_, 1);                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                             \
          }                                                                                                \
        }                                                                                                  \
      }                                                                                                    \
    }

  // ---- fragment addresses (bytes): token row r = lg * 8 + li / 4 of a k-step, 8-byte piece li % 4 of the physical
  // segment (logical ^ ff), ff = f(r) = (li / 4) | ((lg & 1) << 2)
  const int li = lane & 15, lg = lane >> 4;
  const int fr = lg * 8 + (li >> 2);
  const int ff = (li >> 2) | ((lg & 1) << 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  uint32_t aoff[4], boff[2];
#pragma unroll
  for (int x = 0; x < 4; ++x) aoff[x] = lds0 + fr * 256 + (((wr * 4 + x) ^ ff) << 5) + ((li & 3) << 3);
#pragma unroll
  for (int s = 0; s < 2; ++s) boff[s] = lds0 + 2 * QB + fr * 256 + (((wc * 2 + s) ^ ff) << 5) + ((li & 3) << 3);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (P.dbias != nullptr) && tn == 0 && wc == 0;
  const uint32_t one2 = ones_pair<TE>();

  // ---- prologue: K tile 0 entirely, QA0 / QB0 of K tile 1 (issue order = consumption order)
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
  if (nk > 1) { issue(0, 1); issue(1, 1); }
  wgp_wait_quarters(2 + (nk > 1 ? 2 : 0));           // QA0(0), QB0(0) landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                      // ... everyone's
  if (wr == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind from here on

  WgFrag af[4][2], b0[2][2], b1[2][2];
  int bufd = BUFB;                                   // + BUFB / - BUFB: the address registers hop between the two buffers
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---------------- phase 1: read B0 + A lo; issue QB1(kt+1); MFMA A lo x B0
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wgp_read<0>(b0[s][0], boff[s]);
      wgp_read<KSTEP>(b0[s][1], boff[s]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      wgp_read<0>(af[x][0], aoff[x]);
      wgp_read<KSTEP>(af[x][1], aoff[x]);
    }
    if (more1) issue_early(2, kt + 1);
    wgp_wait_pieces(2 + (more1 ? 6 - LATE : 0));     // QB1(kt) for phase 2
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) { wgp_landed(b0[s][0]); wgp_landed(b0[s][1]); }
#pragma unroll
    for (int x = 0; x < 4; ++x) { wgp_landed(af[x][0]); wgp_landed(af[x][1]); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    W2V2_WGP_MFMA(0, 0, b0, more1, 2, kt + 1)
    __builtin_amdgcn_s_setprio(0);
    if (do_bias) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          union { frag8_t v; uint32_t p[4]; } u;
          u.v = af[x][kk].v;
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[x] = pair_sum_add<TE>(u.p[e], one2, bsum[x]);
        }
    }
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2: read B1; issue QA1(kt+1); MFMA A lo x B1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wgp_read<QB>(b1[s][0], boff[s]);
      wgp_read<QB + KSTEP>(b1[s][1], boff[s]);
    }
    if (more1) issue_early(3, kt + 1);
    wgp_wait_pieces(more1 ? 8 - LATE : 0);           // QA1(kt) for phase 3
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s) { wgp_landed(b1[s][0]); wgp_landed(b1[s][1]); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    W2V2_WGP_MFMA(0, 2, b1, more1, 3, kt + 1)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 3: read A hi; issue QA0(kt+2); MFMA A hi x B1
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      wgp_read<QB>(af[x][0], aoff[x]);
      wgp_read<QB + KSTEP>(af[x][1], aoff[x]);
    }
    if (more2) issue_early(0, kt + 2);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int x = 0; x < 4; ++x) { wgp_landed(af[x][0]); wgp_landed(af[x][1]); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    W2V2_WGP_MFMA(4, 2, b1, more2, 0, kt + 2)
    __builtin_amdgcn_s_setprio(0);
    if (do_bias) {
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          union { frag8_t v; uint32_t p[4]; } u;
          u.v = af[x][kk].v;
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum[4 + x] = pair_sum_add<TE>(u.p[e], one2, bsum[4 + x]);
        }
    }
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 4: (operands in registers); issue QB0(kt+2); MFMA A hi x B0
    if (more2) issue_early(1, kt + 2);
    if (more1) wgp_wait_pieces(4 + (more2 ? 4 - LATE : 0));   // QA0(kt+1), QB0(kt+1) for the next K tile's phase 1
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    W2V2_WGP_MFMA(4, 0, b0, more2, 1, kt + 2)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // the next K tile lives in the other buffer
#pragma unroll
    for (int x = 0; x < 4; ++x) aoff[x] += bufd;
#pragma unroll
    for (int s = 0; s < 2; ++s) boff[s] += bufd;
    bufd = -bufd;
  }
#undef W2V2_WGP_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();         // group 0 catches up: every read of the buffers is done

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float s = bsum[i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const int m = m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + li;
      if (lg == 0 && m < P.n_out) P.dbias[m] = s;
    }
  }
  // coalesced f32 tile store through LDS: pass p = rows m0 + 64 p .. + 63 = fragments (p >> 1) * 4 .. + 3 of the waves
  // with wr == (p & 1); a lane holds 4 consecutive n_in of row li per accumulator
  float* stagef = reinterpret_cast<float*>(smem_raw);
  constexpr int PITCH = BN + 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {             // fully unrolled: the accumulator indices must stay static
    __syncthreads();
    if (wr == (pass & 1)) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[(pass >> 1) * 4 + x][j];
          *reinterpret_cast<float4*>(stagef + (x * 16 + li) * PITCH + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + lg * 4) =
              make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 512 * it;                  // 64 rows x 64 float4 chunks
      const int r = c >> 6, ch = c & 63;
      const int m = m0 + pass * 64 + r, n = n0 + ch * 4;
      if (m < P.n_out && n + 4 <= P.n_in)
        store16_wt(P.dW + (int64_t)m * P.ld_dw + n, *reinterpret_cast<const uint4*>(stagef + r * PITCH + ch * 4));
    }
  }
}

template <typename TE, int LATE>
static void launch_phased(const WgArgs& a, int tiles, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * 4 * 64 * 128 * sizeof(bf16_t);   // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_phased_kernel<TE, LATE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_grouped_phased_kernel<TE, LATE>), dim3(tiles), dim3(512), lds, st, a);
}

void w2v2_launch_wgrad_phased(const WgArgs& a, int dtype, int tiles, int late, hipStream_t st) {
  if (dtype == W2V2_BF16) {
    if (late == 1) launch_phased<bf16_t, 1>(a, tiles, st);
    else if (late == 2) launch_phased<bf16_t, 2>(a, tiles, st);
    else launch_phased<bf16_t, 0>(a, tiles, st);
  } else {
    if (late == 1) launch_phased<f16_t, 1>(a, tiles, st);
    else if (late == 2) launch_phased<f16_t, 2>(a, tiles, st);
    else launch_phased<f16_t, 0>(a, tiles, st);
  }
}
