// gemm.hip -- w2v2_gemm: host dispatch + the generic kernels (see include/w2v2_hip.h, "GEMM"; the two large-tile
// LDS-DMA kernels of the training step live in gemm_ring.hip / gemm_phased.hip, shared code in gemm_common.h).
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
//   f32 path  : gemm_f32.hip (exact f32 on the f32-input MFMA; the round-1 VALU tile kernel was deleted in round 4).
#include "gemm_common.h"

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
    // (inline-assembly pieces, common.h: behind the builtin the compiler put `s_waitcnt vmcnt(0)` in front of the first fragment
    // read of the tile being multiplied -- the tile just issued was waited for before anything overlapped it)
#pragma unroll
    for (int j = 0; j < NA; ++j) w2v2_dma16(ap[j] + kt * 64, ad + j * 8 * 64);
#pragma unroll
    for (int j = 0; j < NB; ++j) w2v2_dma16(bp[j] + kt * 64, bd + j * 8 * 64);
  };
  auto landed = [&]() {                           // this wave's pieces are in; the barrier then makes every wave's visible
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
  w2v2_vmcnt0_visible();                          // (common.h: nothing of the compiler's own left to guard inside the loop)
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) stage(1, kt + 1);
    compute(0);
    landed();
    if (kt + 1 < nk) {
      if (kt + 2 < nk) stage(0, kt + 2);
      compute(1);
      landed();
    }
  }

  TC* Cz = reinterpret_cast<TC*>(g.C) + z0 * g.c_s0 + z1 * g.c_s1;
  const TC* auxz = g.aux ? reinterpret_cast<const TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  TC* auxo = g.aux ? reinterpret_cast<TC*>(g.aux) + z0 * g.aux_s0 + z1 * g.aux_s1 : nullptr;
  const float* bias = g.bias ? g.bias + z1 * g.bias_s1 : nullptr;
  tile_epilogue<TC, FM, FN>(g, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, z0, z1, split);
}

static const bool g_w2v2_no_glds = getenv("W2V2_NO_GLDS") != nullptr;   // A/B switches for benchmarking
static const bool g_w2v2_glds3 = getenv("W2V2_NO_GLDS3") == nullptr;
static const bool g_w2v2_tile256 = getenv("W2V2_NO_GEMM_PH") == nullptr;        // 256x256 tiles at all
static const bool g_w2v2_persistent = getenv("W2V2_G3_NONPERSISTENT") == nullptr;
static const int g_w2v2_g3n = getenv("W2V2_G3N") ? atoi(getenv("W2V2_G3N")) : 512;  // smallest N of the 256x128 kernel

static int g_w2v2_ncu = 0;
int w2v2_gemm_device_cus() {
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
//   0 = dispatch below, 1 = 128x128 LDS-DMA, 2 = 256x128 ring, 4 = 256x256x64 phased
// (3 and 5 were the 256x256x32 ring and the 4-wave register-staged kernel: both lost every A/B against the phased
// kernel over two rounds -- DESIGN.md section 4 keeps the numbers -- and were deleted in round 4)
static int g_w2v2_force = 0;
extern "C" int w2v2_tune_gemm_kernel(int family) {
  const int old = g_w2v2_force;
  if (family == 0 || family == 1 || family == 2 || family == 4) g_w2v2_force = family;
  return old;
}

// ------------------------------------------------------------------------------ timed launches (bench.py roofline)
#include <vector>
W2v2PendingTimer& w2v2_pending_timer() {
  static thread_local W2v2PendingTimer t = {nullptr, nullptr, false};
  return t;
}
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_timer_slots;
extern "C" int w2v2_gemm(const w2v2_gemm_desc* d, void* stream);
// Which kernel family w2v2_gemm sends a descriptor to, decided by the dispatch code ITSELF (a dry run: validation +
// decision, no launch): 4 = phased 256x256, 2 = 256x128 ring, 1 = 128-row LDS-DMA kernel, 3 = register-staged, 9 = exact
// f32; < 0 = the descriptor is rejected.  Profilers label launches with it instead of mirroring the heuristics.
static thread_local int g_gemm_dry = 0, g_gemm_family = 0;
extern "C" int w2v2_gemm_kernel_of(const w2v2_gemm_desc* d) {
  g_gemm_dry = 1;
  g_gemm_family = 0;
  const int rc = w2v2_gemm(d, nullptr);
  g_gemm_dry = 0;
  return rc != 0 ? -1 : g_gemm_family;
}
extern "C" int w2v2_gemm_timed(const w2v2_gemm_desc* d, void* stream, int slot) {
  W2V2_REQUIRE(slot >= 0 && slot < (1 << 16), "w2v2_gemm_timed: slot %d out of range", slot);
  while ((int)g_timer_slots.size() <= slot) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) W2V2_FAIL("w2v2_gemm_timed: hipEventCreate failed");
    g_timer_slots.push_back({a, b});
  }
  W2v2PendingTimer& pt = w2v2_pending_timer();
  pt.start = g_timer_slots[slot].first;
  pt.stop = g_timer_slots[slot].second;
  pt.armed = true;
  const int rc = w2v2_gemm(d, stream);
  const bool unused = pt.armed;
  pt.armed = false;
  if (rc != 0) return rc;      // (w2v2_gemm's own message stays in w2v2_last_error)
  if (unused)                  // the product went to a kernel family without the hook: nothing was timed
    W2V2_FAIL("w2v2_gemm_timed: this product does not run on the 256x128 ring or the phased 256x256 kernel");
  return 0;
}
extern "C" int w2v2_timer_read(int first_slot, int n, float* ms_out) {
  W2V2_REQUIRE(first_slot >= 0 && n >= 0 && first_slot + n <= (int)g_timer_slots.size() && ms_out, "w2v2_timer_read: bad range");
  for (int i = 0; i < n; ++i) {
    auto& e = g_timer_slots[first_slot + i];
    if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(ms_out + i, e.first, e.second) != hipSuccess)
      W2V2_FAIL("w2v2_timer_read: slot %d was never launched", first_slot + i);
  }
  return 0;
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
  a.wt_stores = 1;
  a.late_dma = 0;
  a.xcd_tiles = 0;
  const int cal = d->dtype_c == W2V2_F32 ? 4 : 8;     // elements per 16 bytes
  a.c_vec_ok = aligned16(d->C) && (d->ldc % cal == 0) && (d->c_stride0 % cal == 0) && (d->c_stride1 % cal == 0);
  a.aux_vec_ok = d->aux && aligned16(d->aux) && (d->ldaux % cal == 0) && (d->aux_stride0 % cal == 0) &&
                 (d->aux_stride1 % cal == 0);
  hipStream_t st = as_stream(stream);

  if (d->dtype_ab != W2V2_F32) {
    const bool glds = !a.A.trans && !a.B.trans && a.A.vec_ok && a.B.vec_ok && (d->K % 64 == 0) && d->K >= 64 &&
                      !g_w2v2_no_glds;
    // 64-column tiles for N <= 64 -- and, on the LDS-DMA kernel, wherever 128-column tiles would leave more than half of
    // the CUs without a workgroup (the AAM logits: 66 x 5994 x 1536 is 47 tiles of 128 x 128 on 256 CUs streaming 18 MB
    // of class weights: 33.6 -> 29.3 us; the register-staged kernel measured 2 us slower with the narrower tile)
    const bool narrow = d->N <= 64 || (glds && cdiv(d->M, 128) * cdiv(d->N, 128) * split * d->batch * 2 <=
                                                   (int64_t)w2v2_gemm_device_cus());
    const int BM = 128, BN = narrow ? 64 : 128;
    a.tiles_m = (int)cdiv(d->M, BM); a.tiles_n = (int)cdiv(d->N, BN);
    a.k_per_split = (int)(cdiv(cdiv(d->K, split), 64) * 64);
    if (a.k_per_split == 0) a.k_per_split = 64;
    dim3 grid(a.tiles_m * a.tiles_n, split, d->batch);
    // 256x128 3-stage kernel for the encoder shapes; the conv stack (N = 512, M ~ 3e5) measures
    // slightly faster on the 128x128 kernel at 2 workgroups per CU
    // (two-term weights exist on this kernel only: such a request takes it whatever the shape)
    const bool big = glds && split == 1 && !atomic &&
                     (d->k_ext != 0 || (d->N >= g_w2v2_g3n && d->M >= 1024 && g_w2v2_glds3));
    // 256x256 tiles when they fill the chip: >= 85 % of the CU slots of the last round busy (FFN1, dH, conv stack)
    bool huge = false;
    if (big && d->k_ext == 0 && d->N >= 512 && d->batch == 1 && g_w2v2_tile256) {
      const int64_t t4 = cdiv(d->M, 256) * cdiv(d->N, 256), ncu = w2v2_gemm_device_cus();
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
             (double)d->N / (double)(cdiv(d->N, 256) * 256) >= 0.9 && fits32;
    }
    bool big_ = big;
    if (g_w2v2_force != 0 && glds && split == 1 && !atomic && d->k_ext == 0 && d->batch == 1) {
      auto extent = [](const w2v2_operand& o, int64_t rows, int64_t K) -> int64_t {
        return (o.seg_len > 0 ? (rows / o.seg_len + 1) * o.seg_stride + o.seg_len * o.ld : rows * o.ld) + K;
      };
      big_ = g_w2v2_force >= 2;
      huge = g_w2v2_force == 4 && extent(d->A, d->M, d->K) < (int64_t(1) << 31) && extent(d->B, d->N, d->K) < (int64_t(1) << 31) &&
             (d->dtype_c == W2V2_F32 || (a.c_vec_ok && (d->N % 64 == 0) && (d->aux == nullptr || a.aux_vec_ok)));
    }
    if (d->k_ext != 0)
      W2V2_REQUIRE(big && !huge && d->k_ext == d->K && d->n_ext_from >= 0 && d->n_ext_from % 128 == 0 &&
                       d->b_lo_offset % 8 == 0,
                   "w2v2_gemm: two-term weights (k_ext) need k_ext == K, n_ext_from %% 128 == 0 and K-contiguous 16-byte "
                   "aligned operands with K %% 64 == 0 (M=%d N=%d K=%d)", d->M, d->N, d->K);
    // TE = operand element type (selects the MFMA instruction), TC = float or TE
    // Store flavour of the full-line register epilogues: write-through (sc1) stores leave nothing dirty in L2 at the
    // kernel's end (round 3: -0.4 % step time over all writers).  Re-measured per product in round 4
    // (tools/gemm_attrib.py variants 32 / 64, profiles/r04_gemm_attrib.txt): with the ten-instruction IEEE division gone
    // from the GELU epilogue, write-through wins or ties on every product of the step (FFN1 68.8 vs 76.3 us plain, dH 60.0
    // vs 64.2, conv1 568 vs 562), and in-step 12.29 vs 12.48 ms with plain stores everywhere.  W2V2_EPI_WT=0 forces plain
    // write-back stores (A/B).
    {
      static const char* wt_env = getenv("W2V2_EPI_WT");
      a.wt_stores = wt_env ? (wt_env[0] != '0') : 1;
    }
    // Phased kernel, placement of the DMA pieces: in a phase's read segment (the product) or between its MFMAs
    // (W2V2_PH_LATE=1, tools only).  Stand-alone, back to back, the heavy-epilogue products look faster with the pieces
    // between the MFMAs (FFN1 80.3 -> 69.3 us, dH 67.4 -> 63.8; the main loop itself is unchanged, "-epi" columns of
    // profiles/r04_gemm_attrib.txt) -- IN THE STEP the same choice measured +0.6 % (11.50 -> 11.57 ms, three ABAB rounds
    // in one call): an artefact of timing one kernel against itself, not a property of the kernel.  Off.
    {
      static const char* late_env = getenv("W2V2_PH_LATE");
      a.late_dma = late_env ? (late_env[0] != '0') : 0;
    }
    g_gemm_family = huge ? 4 : big_ ? 2 : glds ? 1 : 3;
    if (g_gemm_dry) return 0;                     // w2v2_gemm_kernel_of: the dispatch decision only, nothing launched
#define W2V2_GEMM_LAUNCH(TE, TC)                                                                         \
    do {                                                                                                 \
      if (huge) w2v2_launch_phased_256x256(a, d->dtype_ab, d->dtype_c, d->M, d->N, d->batch, st);        \
      else if (big_) w2v2_launch_ring_256x128(a, d->dtype_ab, d->dtype_c, d->M, d->N, d->batch, g_w2v2_persistent, st); \
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
    // f32 rows are 16-byte vectors of FOUR elements
    auto vec4 = [&](const w2v2_operand& o) {
      return aligned16(o.ptr) && (o.ld % 4 == 0) && (o.seg_stride % 4 == 0) && (o.stride0 % 4 == 0) && (o.stride1 % 4 == 0);
    };
    a.A.vec_ok = vec4(d->A); a.B.vec_ok = vec4(d->B);
    g_gemm_family = w2v2_gemm_f32_dma_rows(a, d->M, d->N, d->K, split, d->batch) > 0 ? 10 : 9;   // 10: gemm_f32_dma.hip
    if (g_gemm_dry) return 0;
    w2v2_launch_gemm_f32(a, d->M, d->N, d->K, split, d->batch, st);       // gemm_f32.hip
  }
  W2V2_CHECK_LAUNCH("w2v2_gemm");
  return 0;
}
