// gemm_f32.hip -- exact-f32 GEMM on the matrix cores (the exact-f32 parity mode of the wav2vec2 engine and BASELINE
// configs[4]: ECAPA-TDNN at the reference's `precision: 32`, "MFMA off" = no reduced-precision matrix path: the numerics
// ARE f32).  See gemm_common.h for the family map; w2v2_gemm (gemm.hip) sends every f32-operand descriptor here.
//
// v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate, bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md 3,
// "FP32-input MFMA") at the f32 vector RATE (157 TFLOP/s) -- issued by one instruction per 4096 multiply-adds instead of
// 64, with one VGPR per operand.
//   BM x BN x 32 block tile, BM, BN in {128, 64}: 4 waves as 2 x 2, (BM/2) x (BN/2) per wave = FI x FJ MFMA blocks of
//   32 x 32.  Both operands are staged K-MAJOR in LDS ([k][row]: a fragment is 32 consecutive rows of one k ->
//   conflict-free ds_read_b32), double-buffered; the next K tile's global loads are issued before the 16 k-steps of the
//   current one and stored to the other buffer behind them.  Operands swapped (D[n][m]) so a lane holds 4 consecutive n
//   per accumulator quad.
//   Tile choice (host): the largest tile whose grid still fills the 512 workgroup slots (2 per CU) -- the 128-channel
//   Res2Net convolutions of ECAPA (N = 128, 3-tile weight gradients) ran 128 x 128 tiles on 30 % of the chip.
//   BK = 32: a K-contiguous operand row contributes one whole 128-byte line per K tile.  LDS pitch: BM + 4 words for
//   K-major sources (16-byte aligned float4 stores), BM + 1 for K-contiguous ones (their transposing scalar stores hit
//   (k + row) % 32 -> 2-way instead of 4-way conflicts); fragment reads are conflict-free with either.
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// (round 5 tried a single-buffered image, 33 KiB = 3 workgroups per CU, two barriers per K tile: 20.40 vs 20.32 ms over the
// ECAPA step's products -- no gain, removed)
template <bool TA, bool TB, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(const GemmArgs g) {
  constexpr int NBUF = 2;
  constexpr int BK = 32, PA = BM + (TA ? 4 : 1), PB = BN + (TB ? 4 : 1);
  constexpr int FI = BM / 64, FJ = BN / 64;          // 32 x 32 MFMA blocks per wave
  constexpr int NA = BM / 32, NB = BN / 32;          // float4 per thread and operand tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float (*As)[BK][PA] = reinterpret_cast<float (*)[BK][PA]>(smem_raw);
  float (*Bs)[BK][PB] = reinterpret_cast<float (*)[BK][PB]>(smem_raw + sizeof(float) * NBUF * BK * PA);
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

  // staging map: NV x float4 per operand and thread.  K-contiguous operand (trans = 0): thread -> (row = c >> 3, 4 k):
  // eight lanes read one 128-byte line; K-major operand (trans = 1): thread -> (k = c / (R / 4), 4 rows)
  auto load_vec = [&](const OpDev& o, const float* __restrict__ base, auto trans_c, auto rows_c, int r0, int rbound, int k0,
                      int j) -> float4 {
    constexpr bool trans = decltype(trans_c)::value;
    constexpr int R = decltype(rows_c)::value;
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
      const int k = k0 + c / (R / 4), row = (c % (R / 4)) * 4;
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
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  auto store_a1 = [&](int buf, float4 r, int j) {
    const int c = tid + 256 * j;
    if constexpr (!TA) {
      const int row = c >> 3, k = (c & 7) * 4;
      As[buf][k][row] = r.x; As[buf][k + 1][row] = r.y; As[buf][k + 2][row] = r.z; As[buf][k + 3][row] = r.w;
    } else {
      *reinterpret_cast<float4*>(&As[buf][c / (BM / 4)][(c % (BM / 4)) * 4]) = r;
    }
  };
  auto store_b1 = [&](int buf, float4 r, int j) {
    const int c = tid + 256 * j;
    if constexpr (!TB) {
      const int row = c >> 3, k = (c & 7) * 4;
      Bs[buf][k][row] = r.x; Bs[buf][k + 1][row] = r.y; Bs[buf][k + 2][row] = r.z; Bs[buf][k + 3][row] = r.w;
    } else {
      *reinterpret_cast<float4*>(&Bs[buf][c / (BN / 4)][(c % (BN / 4)) * 4]) = r;
    }
  };

  f32x16 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  // Interior tiles of plain (unsegmented, 16-byte aligned) operands: the per-thread source pointers are formed ONCE and
  // advance by a constant per K tile.  The general load_vec above re-derives row offsets, bounds and the vector / scalar
  // choice for each of its 8 loads in every K tile (~500 mostly scalar instructions in front of the 64 MFMAs of a wave,
  // round 5: the in-order wave cannot issue MFMAs meanwhile); it stays for edge tiles, segmented operands and K tails.
  const bool fast_tile = g.A.vec_ok && g.B.vec_ok && g.A.seg_len <= 0 && g.B.seg_len <= 0 && m0 + BM <= g.M &&
                         n0 + BN <= g.N;
  const float* pa[NA];
  const float* pb[NB];
  int64_t adv_a = 0, adv_b = 0;
  if (fast_tile) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int c = tid + 256 * j;
      if constexpr (!TA) pa[j] = Ab + (int64_t)(m0 + (c >> 3)) * g.A.ld + kbeg + (c & 7) * 4;
      else pa[j] = Ab + (int64_t)(kbeg + c / (BM / 4)) * g.A.ld + m0 + (c % (BM / 4)) * 4;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int c = tid + 256 * j;
      if constexpr (!TB) pb[j] = Bb + (int64_t)(n0 + (c >> 3)) * g.B.ld + kbeg + (c & 7) * 4;
      else pb[j] = Bb + (int64_t)(kbeg + c / (BN / 4)) * g.B.ld + n0 + (c % (BN / 4)) * 4;
    }
    adv_a = TA ? (int64_t)BK * g.A.ld : BK;
    adv_b = TB ? (int64_t)BK * g.B.ld : BK;
  }
  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (fast_tile && k0 + BK <= kend) {
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = *reinterpret_cast<const float4*>(pa[j] + (int64_t)kt * adv_a);
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const float4*>(pb[j] + (int64_t)kt * adv_b);
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = load_vec(g.A, Ab, std::integral_constant<bool, TA>{}, std::integral_constant<int, BM>{}, m0, g.M, k0, j);
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = load_vec(g.B, Bb, std::integral_constant<bool, TB>{}, std::integral_constant<int, BN>{}, n0, g.N, k0, j);
    }
  };
  if (nk > 0) {
    load_tile(0);
#pragma unroll
    for (int j = 0; j < NA; ++j) store_a1(0, ra[j], j);
#pragma unroll
    for (int j = 0; j < NB; ++j) store_b1(0, rb[j], j);
  }
  __syncthreads();
  const int kl = lane >> 5, rl = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tile(kt + 1);
    // Fragments of k-step kk + 2 are requested BEFORE the MFMAs of step kk issue (two register sets, the scheduler fenced
    // so that it cannot fold them back into one): left alone the compiler reuses one set, so every step's ds_reads
    // queue behind the previous step's last MFMA issue and their latency sits in front of the next four MFMAs
    float af[2][FI], bf[2][FJ];
    auto frag = [&](int set, int kk) {
#pragma unroll
      for (int i = 0; i < FI; ++i) af[set][i] = As[cur][kk + kl][wm * (BM / 2) + i * 32 + rl];
#pragma unroll
      for (int j = 0; j < FJ; ++j) bf[set][j] = Bs[cur][kk + kl][wn * (BN / 2) + j * 32 + rl];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int set = (kk >> 1) & 1;
      if (kk + 2 < BK) frag(set ^ 1, kk + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j], af[set][i], acc[i][j], 0, 0, 0);   // D[n][m]
      __builtin_amdgcn_sched_barrier(0);
    }
    // (round 4 tried spreading these stores over the second half of the k-steps: the `vmcnt` wait in front of the first
    // one then sits in the MIDDLE of the MFMA stream and stops its issue -- 19800 x 1024 x 1024: 435 -> 518 us; behind the
    // last MFMA the wait overlaps the matrix pipe draining)
    if (more) {
#pragma unroll
      for (int j = 0; j < NA; ++j) store_a1(cur ^ 1, ra[j], j);
#pragma unroll
      for (int j = 0; j < NB; ++j) store_b1(cur ^ 1, rb[j], j);
    }
    __syncthreads();
  }

  float* Cz = reinterpret_cast<float*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const float* auxz = g.aux ? reinterpret_cast<const float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  float* auxo = g.aux ? reinterpret_cast<float*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  // D[n][m]: lane l, register r: m = l & 31, n = 8 (r >> 2) + 4 (l >> 5) + (r & 3)
  if (g.atomic && (g.epilogue == W2V2_EPI_NONE || g.epilogue == W2V2_EPI_BIAS)) {
    // Split-K / accumulating products (the token-long weight gradients of ECAPA: 8..32 partial tiles per output tile)
    // ADD into C.  Straight from the accumulator layout one atomic instruction touches 32 different ROWS = 64 cache
    // lines, and the L2 atomic units then are the bottleneck of the whole launch (round 4, ECAPA step: 1024 x 1024 x
    // 19800 split 8 ways 553 us with these atomics, 392 us with plain stores instead).  So each 32-row block goes
    // through LDS (the operand images are dead) and comes back with the lanes ALONG a row: one instruction = 64 (or
    // 2 x 32) consecutive columns = 4 lines.
    constexpr int WN = BN / 2, SP = WN + 1;
    static_assert(4 * 32 * SP * sizeof(float) <= sizeof(float) * NBUF * 32 * (PA + PB), "staging fits the operand images");
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (32 * SP);
    const bool add_bias = g.epilogue == W2V2_EPI_BIAS && split == 0 && bias != nullptr;
#pragma unroll
    for (int i = 0; i < FI; ++i) {
#pragma unroll
      for (int j = 0; j < FJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[rl * SP + j * 32 + 8 * (r >> 2) + 4 * kl + (r & 3)] = acc[i][j][r] * g.alpha;
      __syncthreads();
      const int mb = m0 + wm * (BM / 2) + i * 32;
      if constexpr (WN == 64) {
        const int n = n0 + wn * WN + lane;
        const float bv = (add_bias && n < g.N) ? bias[n] : 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr)
          if (mb + rr < g.M && n < g.N) unsafeAtomicAdd(Cz + (int64_t)(mb + rr) * g.ldc + n, stage[rr * SP + lane] + bv);
      } else {
        const int n = n0 + wn * WN + (lane & 31);
        const float bv = (add_bias && n < g.N) ? bias[n] : 0.f;
#pragma unroll 8
        for (int r2 = 0; r2 < 16; ++r2) {
          const int rr = 2 * r2 + (lane >> 5);
          if (mb + rr < g.M && n < g.N) unsafeAtomicAdd(Cz + (int64_t)(mb + rr) * g.ldc + n, stage[rr * SP + (lane & 31)] + bv);
        }
      }
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        epilogue_store4<float>(g, Cz, auxz, auxo, bias, m0 + wm * (BM / 2) + i * 32 + rl,
                               n0 + wn * (BN / 2) + j * 32 + q * 8 + kl * 4, v4, split == 0);
      }
}

template <bool TA, bool TB, int BM, int BN>
static void launch_f32(GemmArgs a, int M, int N, int split, int batch, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * 2 * 32 * ((BM + (TA ? 4 : 1)) + (BN + (TB ? 4 : 1)));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<TA, TB, BM, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  a.tiles_m = (int)cdiv(M, BM);
  a.tiles_n = (int)cdiv(N, BN);
  dim3 grid(a.tiles_m * a.tiles_n, split, batch);
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<TA, TB, BM, BN>), grid, dim3(256), lds, st, a);
}

template <int BM, int BN>
static void launch_f32_layout(const GemmArgs& a, int M, int N, int split, int batch, hipStream_t st) {
  if (!a.A.trans && !a.B.trans) launch_f32<false, false, BM, BN>(a, M, N, split, batch, st);
  else if (!a.A.trans && a.B.trans) launch_f32<false, true, BM, BN>(a, M, N, split, batch, st);
  else if (a.A.trans && !a.B.trans) launch_f32<true, false, BM, BN>(a, M, N, split, batch, st);
  else launch_f32<true, true, BM, BN>(a, M, N, split, batch, st);
}

// tools: 0 = the choice below; 1..4 = the register-staged kernel on 128x128 / 64x128 / 128x64 / 64x64; 11..15 = the LDS-DMA
// kernel (gemm_f32_dma.hip) on (32 fi) x 128 tiles, fi = t - 10; + 100: the same with XCD-contiguous tiles (+ 200, 400, 800:
// timing-only variants without the loop's DMA / barrier / vmcnt wait).  An ineligible product ignores a DMA code.
static int g_f32_tile_force = 0;
static int g_f32_last_kernel = 0;
extern "C" int w2v2_gemm_f32_last_kernel(void) { return g_f32_last_kernel; }
extern "C" int w2v2_tune_gemm_f32_tile(int t) {
  const int old = g_f32_tile_force;
  const int b = t % 100;
  if (t >= 0 && t < 1600 && ((b >= 0 && b <= 4) || (b >= 11 && b <= 15))) g_f32_tile_force = t;
  return old;
}

// Which kernel an exact-f32 product runs on: 0 = the register-staged kernel below, else the tile height fi (rows = 32 fi)
// of the LDS-DMA kernel (gemm_f32_dma.hip).  Also asked by w2v2_gemm's dry run (w2v2_gemm_kernel_of: family 10 instead of 9).
int w2v2_gemm_f32_dma_rows(const GemmArgs& a, int M, int N, int K, int split, int batch) {
  // plain 16-byte aligned operands whose every 16-byte piece is whole
  static const bool no_dma = [] { const char* e = getenv("W2V2_F32_NO_DMA"); return e && e[0] != '0'; }();
  const bool dma_ok = !no_dma && a.A.vec_ok && a.B.vec_ok && a.A.seg_len <= 0 && a.B.seg_len <= 0 && K % 4 == 0 &&
                      (!a.A.trans || M % 4 == 0) && (!a.B.trans || N % 4 == 0) && M > 64 && N > 64;
  if (!dma_ok) return 0;
  const int force = g_f32_tile_force % 100;
  if (force >= 11 && force <= 15) return force - 10;
  if (force != 0) return 0;
  // Rows per tile.  The matrix pipes of a CU are shared by its resident workgroups, so what a grid quantises over is
  // TILES PER CU, not workgroup slots: cost = ceil(tiles / CUs) x (fi + 0.6), the 0.6 standing for a tile's
  // prologue + epilogue.  A lone workgroup on a CU cannot hide its own barriers and fragment reads (x 1.35).  Short
  // K ranges (<= 16 K tiles) are bound by the latency of the first tiles, which only more resident workgroups hide:
  // fi = 1 (40 KiB of LDS, four per CU).  From every product of the ECAPA step on every fi
  // (tools/f32_dma_sweep.sh, profiles/r06_f32_dma_sweep.txt).
  const int64_t cus = w2v2_gemm_device_cus();
  const int64_t kper = cdiv(K, split);
  if (kper <= 512) return 1;
  int fi = 1;
  double best = 1e30;
  for (int f = 1; f <= 5; ++f) {
    const int64_t w = cdiv(M, 32 * f) * cdiv(N, 128) * (int64_t)split * batch;
    const int64_t per_cu = cdiv(w, cus);
    const double cost = (double)per_cu * (f + 0.6) * (per_cu == 1 ? 1.35 : 1.0);
    if (cost <= best) { best = cost; fi = f; }
  }
  return fi;
}

void w2v2_launch_gemm_f32(GemmArgs a, int M, int N, int K, int split, int batch, hipStream_t st) {
  a.k_per_split = (int)(cdiv(cdiv(K, split), 32) * 32);
  if (a.k_per_split == 0) a.k_per_split = 32;
  // Tile choice, from every product of the ECAPA step timed on each tile (tools/ecapa_gemm_shapes.py F32_TILE=1..4,
  // round 5: best-per-shape 19.3 ms against 20.2 for the round-4 rule "largest tile that fills 60 % of the slots"):
  //   * 128 x 128 (2 workgroups per CU = 512 slots, the fastest main loop) when the grid is at least ~0.9 of a round AND
  //     its last round is not too empty (rounds / exact rounds <= 1.12: 19800 x 1024 x 1024 = 2.42 rounds loses 19 % to
  //     the third round and runs faster on 64 x 64 tiles, 4 per CU, 4.84 rounds) AND the K loop is long enough to pay
  //     for the big tile's prologue and epilogue (K > 256);
  //   * else the 64-row / 64-column tile along the shorter side if THAT fills its 768 slots the same way;
  //   * else 64 x 64: skinny and short-K products (N = 128 Res2Net convolutions, K = 128 .. 200) are bound by the
  //     global-load latency of a 12-trip K loop, which only more resident workgroups hide (41.7 -> 30.6 us).
  const int64_t cus = w2v2_gemm_device_cus();
  const int fi = w2v2_gemm_f32_dma_rows(a, M, N, K, split, batch);
  if (fi > 0) {
    a.xcd_tiles = g_f32_tile_force / 100;         // bit 0: XCD-contiguous tiles; bits 1-3: timing-only variants (tools)
    g_f32_last_kernel = 10 * fi + 2;
    w2v2_launch_gemm_f32_dma(a, M, N, split, batch, fi, 2, st);
    return;
  }
  g_f32_last_kernel = 0;
  auto wgs = [&](int bm, int bn) { return cdiv(M, bm) * cdiv(N, bn) * (int64_t)split * batch; };
  auto fits = [&](int bm, int bn, int per_cu) {
    const double exact = (double)wgs(bm, bn) / (double)(per_cu * cus);
    return exact >= 0.9 && (double)cdiv(wgs(bm, bn), per_cu * cus) / exact <= 1.12;
  };
  int bm = 64, bn = 64;
  const int64_t kper = cdiv(K, split);
  if (M > 64 && N > 64 && kper > 256) {
    if (fits(128, 128, 2)) { bm = 128; bn = 128; }
    else if (M >= N && fits(64, 128, 3)) { bm = 64; bn = 128; }
    else if (M < N && fits(128, 64, 3)) { bm = 128; bn = 64; }
  }
  switch (g_f32_tile_force) {
    case 1: bm = 128; bn = 128; break;
    case 2: bm = 64; bn = 128; break;
    case 3: bm = 128; bn = 64; break;
    case 4: bm = 64; bn = 64; break;
    default: break;
  }
  if (bm == 128 && bn == 128) launch_f32_layout<128, 128>(a, M, N, split, batch, st);
  else if (bm == 64 && bn == 128) launch_f32_layout<64, 128>(a, M, N, split, batch, st);
  else if (bm == 128 && bn == 64) launch_f32_layout<128, 64>(a, M, N, split, batch, st);
  else launch_f32_layout<64, 64>(a, M, N, split, batch, st);
}
