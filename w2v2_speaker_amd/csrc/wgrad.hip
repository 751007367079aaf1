// wgrad.hip -- grouped weight-gradient GEMM:  dW_p[o][i] = sum_t dY_p[t][o] * X_p[t][i]   (+ dbias_p[o] = sum_t dY_p[t][o])
// for up to 8 (dY, X) pairs in ONE launch (the four Linear layers of a transformer block: QKV, out-proj,
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
#include "common.cuh"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) short4v lds_s4_t;

constexpr int WG_MAXP = 8;

struct WgProblem {
  const bf16_t* dY;
  const bf16_t* X;
  float* dW;
  float* dbias;
  int64_t ld_dy, ld_x, ld_dw;
  int n_out, n_in;
  int tile_begin, tiles_n;
};
struct WgArgs {
  WgProblem p[WG_MAXP];
  int n_problems, total_tiles, ktiles;
};

__device__ __forceinline__ int wg_f(int r) { return (r & 3) | ((r >> 1) & 4); }

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int kk, int seg, int lane) {
  // k-slots of lane group g: rows kk*32 + g*8 + {0..7}; two 4x16 transposing reads
  const int i = lane & 15, g = lane >> 4;
  const int r = kk * 32 + g * 8 + (i >> 2);
  const int f = (i >> 2) | ((g & 1) << 2);
  const bf16_t* p = tile + r * 128 + ((seg ^ f) << 4) + ((i & 3) << 2);
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)p);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(p + 4 * 128));
  union { struct { short4v a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

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
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(Ac, kk, wm * 4 + i, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(Bc, kk, wn * 4 + j, lane);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[i] += (float)af[i][e];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
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
        *reinterpret_cast<float4*>(P.dW + (int64_t)m * P.ld_dw + n) =
            *reinterpret_cast<const float4*>(stagef + r * PITCH + ch * 4);
    }
  }
}

extern "C" int w2v2_wgrad_grouped(const w2v2_wgrad_problem* probs, int n, int tokens, int tokens_padded, void* stream) {
  W2V2_REQUIRE(probs && n > 0 && n <= WG_MAXP, "wgrad_grouped: need 1..%d problems", WG_MAXP);
  W2V2_REQUIRE(tokens > 0 && tokens_padded >= tokens && tokens_padded % 64 == 0,
               "wgrad_grouped: tokens_padded must be tokens rounded up to a multiple of 64");
  WgArgs a;
  int tiles = 0;
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
    P.tiles_n = (int)cdiv(q.n_in, 128);
    P.tile_begin = tiles;
    tiles += (int)cdiv(q.n_out, 128) * P.tiles_n;
  }
  for (int i = n; i < WG_MAXP; ++i) a.p[i] = a.p[0];
  a.n_problems = n;
  a.total_tiles = tiles;
  a.ktiles = tokens_padded / 64;
  constexpr size_t lds = (size_t)2 * 2 * 64 * 128 * sizeof(bf16_t);   // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(tiles), dim3(256), lds, as_stream(stream), a);
  W2V2_CHECK_LAUNCH("wgrad_grouped");
  return 0;
}
