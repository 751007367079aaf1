// attention.hip -- fused multi-head self-attention for training-length sequences (T <= 256,
// head dim 64): HF:438-548 minus the q/k/v/out projections.  bf16 in, f32 softmax, bf16 out.
//
// All three kernels share one skeleton.  A wave owns 16 "rows" (queries in fwd / dQ, keys in dK/dV)
// whose operand fragments live in registers; the whole "column" matrix of the (batch, head) sits in
// LDS twice: row-major [n][64] (XOR-swizzled 16-B chunks, read as ds_read_b128 MFMA fragments) for
// the score-like products, and transposed [64][n] (read as 2 x ds_read_b64) for the product that
// contracts over n.  Scores are computed with swapped MFMA operands (D[row=n][col=m]) so each lane
// holds, for ONE of its 16 rows, 4 consecutive columns per 16x16 fragment: the softmax row
// reductions are in-register + two wave shuffles (xor 16, 32), and the probabilities feed the second
// MFMA directly from registers (never through LDS or HBM).  The MFMA k-slot <-> column mapping
// (slot (g,e): column blk*32 + (e<4 ? g*4+e : 16+g*4+e-4)) is applied identically to the register
// operand and the transposed-LDS operand, which is all a contraction needs.
//
//   fwd   : S = QK^T*scale -> P = softmax(S) (+dropout) -> O = P V            saves LSE[b,h,q]
//   bwd_dq: recompute P; dP = dO V^T; dS = P*(dP - delta)*scale; dQ = dS K    writes delta[b,h,q]
//   bwd_kv: (rows = keys) recompute P^T; dV = Pdrop^T dO; dK = dS^T Q
// Backward recomputes the scores in both kernels (7 matmul units instead of 5) in exchange for
// no [B,h,T,T] tensor in HBM at all; attention is 3 % of the step's FLOPs.
#include "common.cuh"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int HD = 64;  // head dim

// 16-row periodic: all fragments of an image share one per-lane swizzle, so fragment addresses are
// lane base + compile-time constant (ds_read offset immediates instead of one address VGPR each)
// dropout counter of attention probability (bh, q, key): rows are padded to an even length so that keys 2j, 2j+1
// of a row always share one hash (common.cuh rng_pair); every fused kernel (forward, all backward variants) uses it
__device__ __forceinline__ uint64_t attn_drop_idx(int64_t bh, int q, int key, int Tn) {
  return (uint64_t)(bh * Tn + q) * (uint64_t)((Tn + 1) & ~1) + (uint64_t)key;
}
// keep-scales of 4 consecutive keys key0 .. key0+3 (key0 % 4 == 0) of query row q: two hashes
__device__ __forceinline__ void attn_drop4_keys(uint64_t seed, int64_t bh, int q, int key0, int Tn, float dp,
                                                float inv_keep, float (&ms)[4]) {
  const uint32_t k = rng_key(seed), thr = drop_thr16(dp);
  const uint64_t pi = attn_drop_idx(bh, q, key0, Tn) >> 1;
  const uint32_t h0 = rng_pair(k, pi), h1 = rng_pair(k, pi + 1);
  ms[0] = (h0 & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[1] = (h0 >> 16) >= thr ? inv_keep : 0.f;
  ms[2] = (h1 & 0xffffu) >= thr ? inv_keep : 0.f;
  ms[3] = (h1 >> 16) >= thr ? inv_keep : 0.f;
}
// keep-scales of key column `key` (parity == lane parity) for 4 consecutive query rows q0 .. q0+3: the two lanes
// of a key pair hash two rows each and swap (one hash serves keys 2j and 2j+1 of a row)
__device__ __forceinline__ void attn_drop4_rows(uint64_t seed, int64_t bh, int q0, int key, int Tn, float dp,
                                                float inv_keep, float (&ms)[4]) {
  const uint32_t k = rng_key(seed), thr = drop_thr16(dp);
  const int odd = key & 1;
  const uint32_t h0 = rng_pair(k, attn_drop_idx(bh, q0 + 2 * odd, key, Tn) >> 1);
  const uint32_t h1 = rng_pair(k, attn_drop_idx(bh, q0 + 2 * odd + 1, key, Tn) >> 1);
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
// MFMA fragment from the transposed image: row d0 + lane&15, k-slots of 32-column block `blk`
__device__ __forceinline__ frag8_t lds_frag_t(const bf16_t* ldst, int pitch, int d0, int blk, int lane) {
  const bf16_t* p = ldst + (d0 + (lane & 15)) * pitch + blk * 32 + (lane >> 4) * 4;
  const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 16);
  return as_frag(make_uint4(a.x, a.y, b.x, b.y));
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
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ===================================================================================== v2: one workgroup per (b, h)
// For T <= 160 (the 3 s training clips: T = 149 / 150) a single workgroup owns a whole (batch, head):
// K/V/Q/dO are fetched ONCE, with all global loads of a thread issued before the first LDS store (the
// v1 fill loops were a chain of dependent load->store round trips, ~18 us per workgroup), and both the
// row-major and the transposed LDS images are produced from the same registers.  The backward is one
// kernel: phase A (waves own query fragments) produces dQ and delta, phase B (waves own key fragments)
// produces dK and dV from the same LDS images.
template <int NF> struct BlkRegs {
  static constexpr int NIT = (NF * 16 / 4 * 8 + 255) / 256;   // 4-row x 8-col micro-blocks per thread
  uint4 v[NIT][4];
};

template <int NF>
__device__ __forceinline__ void blk_load(BlkRegs<NF>& r, const bf16_t* g, int64_t gs, int n_valid) {
  constexpr int NB = NF * 16 / 4 * 8;
#pragma unroll
  for (int it = 0; it < BlkRegs<NF>::NIT; ++it) {
    const int blk = threadIdx.x + 256 * it;
    const int cb = blk & 7, rb = blk >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rb * 4 + i;
      r.v[it][i] = make_uint4(0, 0, 0, 0);
      if (blk < NB && row < n_valid) r.v[it][i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * gs + cb * 8);
    }
  }
}
template <int NF>
__device__ __forceinline__ void blk_store_rows(const BlkRegs<NF>& r, bf16_t* lds) {
  constexpr int NB = NF * 16 / 4 * 8;
#pragma unroll
  for (int it = 0; it < BlkRegs<NF>::NIT; ++it) {
    const int blk = threadIdx.x + 256 * it;
    if (blk >= NB) continue;
    const int cb = blk & 7, rb = blk >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rb * 4 + i;
      *reinterpret_cast<uint4*>(lds + row * 64 + ((cb ^ aswz(row)) << 3)) = r.v[it][i];
    }
  }
}
template <int NF>
__device__ __forceinline__ void blk_store_t(const BlkRegs<NF>& r, bf16_t* ldst, int pitch) {
  constexpr int NB = NF * 16 / 4 * 8;
#pragma unroll
  for (int it = 0; it < BlkRegs<NF>::NIT; ++it) {
    const int blk = threadIdx.x + 256 * it;
    if (blk >= NB) continue;
    const int cb = blk & 7, rb = blk >> 3;
    const uint32_t w[4][4] = {{r.v[it][0].x, r.v[it][0].y, r.v[it][0].z, r.v[it][0].w},
                              {r.v[it][1].x, r.v[it][1].y, r.v[it][1].z, r.v[it][1].w},
                              {r.v[it][2].x, r.v[it][2].y, r.v[it][2].z, r.v[it][2].w},
                              {r.v[it][3].x, r.v[it][3].y, r.v[it][3].z, r.v[it][3].w}};
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
      const int d = ci >> 1;
      uint2 o;
      if (ci & 1) {
        o.x = (w[0][d] >> 16) | (w[1][d] & 0xffff0000u);
        o.y = (w[2][d] >> 16) | (w[3][d] & 0xffff0000u);
      } else {
        o.x = (w[0][d] & 0xffffu) | (w[1][d] << 16);
        o.y = (w[2][d] & 0xffffu) | (w[3][d] << 16);
      }
      *reinterpret_cast<uint2*>(ldst + (cb * 8 + ci) * pitch + rb * 4) = o;
    }
  }
}
template <typename TE>
__device__ __forceinline__ void store_row4x4(bf16_t* dst, const f32x4 (&o)[4]) {
#pragma unroll
  for (int df = 0; df < 4; ++df) {
    uint2 w;
    w.x = pack2<TE>(o[df][0], o[df][1]);
    w.y = pack2<TE>(o[df][2], o[df][3]);
    *reinterpret_cast<uint2*>(dst + df * 16) = w;
  }
}

template <typename TE, int NF>
__global__ __launch_bounds__(256) void attn_fwd2_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                        float* __restrict__ lse, int Tn, int heads, float scale,
                                                        float dp, float inv_keep, uint64_t seed) {
  constexpr int TP = NF * 16, PITCH = TP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);   // [TP][64]
  bf16_t* Qs = Ks + TP * 64;                      // [TP][64]
  bf16_t* Vt = Qs + TP * 64;                      // [64][PITCH]
  const int b = blockIdx.y, h = blockIdx.x;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  {
    BlkRegs<NF> rq, rk, rv;
    blk_load<NF>(rq, qb, gs, Tn);
    blk_load<NF>(rk, qb + H, gs, Tn);
    blk_load<NF>(rv, qb + 2 * H, gs, Tn);
    blk_store_rows<NF>(rq, Qs);
    blk_store_rows<NF>(rk, Ks);
    blk_store_t<NF>(rv, Vt, PITCH);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int64_t bh = (int64_t)b * heads + h;
#pragma unroll 1
  for (int qf = wave; qf < NF; qf += 4) {
    asm volatile("" ::: "memory");   // keep the K / V^T fragment loads inside the loop (LICM would hoist 240 VGPRs)
    const int q = qf * 16 + (lane & 15);
    int g4 = g * 4;
    asm volatile("" : "+v"(g4));          // opaque: no hoisting of the 40 per-column index / RNG-counter values
    frag8_t qfr[2] = {lds_frag(Qs, qf * 16, 0, lane), lds_frag(Qs, qf * 16, 1, lane)};
    float s[NF][4];
    float mx = -INFINITY;
#pragma unroll
    for (int fj = 0; fj < NF; ++fj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        acc = mfma16<TE>(lds_frag(Ks, fj * 16, kk, lane), qfr[kk], acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = fj * 16 + g4 + j;
        s[fj][j] = key < Tn ? acc[j] * scale : -INFINITY;
        mx = fmaxf(mx, s[fj][j]);
      }
    }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int fj = 0; fj < NF; ++fj)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[fj][j] = __expf(s[fj][j] - mx); sum += s[fj][j]; }
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
    if (g == 0 && q < Tn) lse[bh * Tn + q] = mx + __logf(sum);
#pragma unroll
    for (int fj = 0; fj < NF; ++fj) {
      float ms[4] = {1.f, 1.f, 1.f, 1.f};
      if (dp > 0.f) attn_drop4_keys(seed, bh, q, fj * 16 + g4, Tn, dp, inv_keep, ms);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[fj][j] = s[fj][j] * inv * ms[j];
    }
    f32x4 o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NF / 2; ++kb) {
      const frag8_t pf = pack_frag<TE>(s[2 * kb], s[2 * kb + 1]);
#pragma unroll
      for (int df = 0; df < 4; ++df)
        o[df] = mfma16<TE>(lds_frag_t(Vt, PITCH, df * 16, kb, lane), pf, o[df]);
    }
    if (q < Tn) store_row4x4<TE>(ctx + ((int64_t)b * Tn + q) * H + h * HD + g * 4, o);
  }
}

// ===================================================================================== v3 backward: tr-read, 2 workgroups / CU
// The merged backward above keeps 4 row-major and 3 transposed LDS images (144 KiB): ONE workgroup of 4 waves
// per CU, i.e. one wave per SIMD and every LDS / MFMA / exp latency exposed.  Here the contraction-side
// operands (K for dQ, dO and Q for dV / dK) are read straight from the ROW-MAJOR images with the hardware
// transpose read ds_read_b64_tr_b16 (16 lanes x 8 B = a 4(row) x 16(col) block, lane i receives column i), so
// only K, V, Q, dO row-major remain: exactly 80 KiB -> two workgroups per CU (8 waves, 2 per SIMD).  dS / P are
// consumed per 32-column block as soon as they exist (no [T] x 4 register arrays), delta goes through a
// global scratch row (written in phase A, read in phase B of the same workgroup).
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

template <typename TE, int NF>
__global__ __launch_bounds__(256, 2) void attn_bwd3_kernel(const bf16_t* __restrict__ qkv,
                                                           const bf16_t* __restrict__ ctx,
                                                           const bf16_t* __restrict__ dctx,
                                                           const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                           float* __restrict__ delta, int Tn, int heads, float scale,
                                                           float dp, float inv_keep, uint64_t seed) {
  constexpr int TP = NF * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);   // row-major swizzled images [TP][64]
  bf16_t* Vs = Ks + TP * 64;
  bf16_t* Qs = Vs + TP * 64;
  bf16_t* Os = Qs + TP * 64;                      // dO
  const int b = blockIdx.y, h = blockIdx.x;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const bf16_t* dob = dctx + (int64_t)b * Tn * H + h * HD;
  const bf16_t* ob = ctx + (int64_t)b * Tn * H + h * HD;
  const int64_t bh = (int64_t)b * heads + h;
  {
    BlkRegs<NF> ra, rb;
    blk_load<NF>(ra, qb, gs, Tn);
    blk_load<NF>(rb, qb + H, gs, Tn);
    blk_store_rows<NF>(ra, Qs);
    blk_store_rows<NF>(rb, Ks);
    blk_load<NF>(ra, qb + 2 * H, gs, Tn);
    blk_load<NF>(rb, dob, H, Tn);
    blk_store_rows<NF>(ra, Vs);
    blk_store_rows<NF>(rb, Os);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const float* lse_b = lse + bh * Tn;
  float* del_b = delta + bh * Tn;
  const TrOff troff = tr_offsets(lane);
  __syncthreads();

  // ---- phase A: waves own query fragments -> dQ, delta
#pragma unroll 1
  for (int qf = wave; qf < NF; qf += 4) {
    asm volatile("" ::: "memory");
    const int q = qf * 16 + (lane & 15);
    frag8_t qfr[2] = {lds_frag(Qs, qf * 16, 0, lane), lds_frag(Qs, qf * 16, 1, lane)};
    frag8_t dof[2] = {lds_frag(Os, qf * 16, 0, lane), lds_frag(Os, qf * 16, 1, lane)};
    frag8_t of[2];
    reg_frag(of, ob, H, q, Tn, lane);
    float dl = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) dl += frag_dot<TE>(dof[kk], of[kk]);
    dl = quad_sum(dl);
    if (g == 0 && q < Tn) del_b[q] = dl;
    const float l = q < Tn ? lse_b[q] : 0.f;
    f32x4 o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NF / 2; ++kb) {
      asm volatile("" ::: "memory");      // bound the scheduler's load hoisting to one 32-key block
      int g4 = g * 4;
      asm volatile("" : "+v"(g4));        // opaque: keeps the per-column index / predicate / RNG-counter math of
                                          // all 40 columns from being hoisted out of the fragment loop (and spilled)
      float ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fj = 2 * kb + hf;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          sa = mfma16<TE>(lds_frag(Ks, fj * 16, kk, lane), qfr[kk], sa);
          pa = mfma16<TE>(lds_frag(Vs, fj * 16, kk, lane), dof[kk], pa);
        }
        float ms[4] = {1.f, 1.f, 1.f, 1.f};
        if (dp > 0.f) attn_drop4_keys(seed, bh, q, fj * 16 + g4, Tn, dp, inv_keep, ms);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = fj * 16 + g4 + j;
          const float p = (key < Tn && q < Tn) ? __expf(sa[j] * scale - l) : 0.f;
          ds2[hf][j] = p * (pa[j] * ms[j] - dl) * scale;
        }
      }
      const frag8_t pf = pack_frag<TE>(ds2[0], ds2[1]);
#pragma unroll
      for (int df = 0; df < 4; ++df)
        o[df] = mfma16<TE>(lds_frag_tr(Ks, troff, df, kb), pf, o[df]);
    }
    if (q < Tn) store_row4x4<TE>(dqkv + ((int64_t)b * Tn + q) * gs + h * HD + g * 4, o);
  }
  __threadfence_block();
  __syncthreads();          // delta of every query row of this (b, h) is visible to the workgroup

  // ---- phase B: waves own key fragments -> dK, dV
  // every wave keeps the (b, h)'s LSE and delta rows distributed over its lanes (3 VGPRs each: row r lives in
  // lane r & 63, register r >> 6) and fetches the 4 values a fragment needs with cross-lane reads
  constexpr int NR = (TP + 63) / 64;
  float lse_r[NR], del_r[NR];
#pragma unroll
  for (int k2 = 0; k2 < NR; ++k2) {
    const int r = k2 * 64 + lane;
    lse_r[k2] = r < Tn ? lse_b[r] : 0.f;
    del_r[k2] = r < Tn ? del_b[r] : 0.f;
  }
#pragma unroll 1
  for (int kf = wave; kf < NF; kf += 4) {
    asm volatile("" ::: "memory");
    const int key = kf * 16 + (lane & 15);
    frag8_t kfr[2] = {lds_frag(Ks, kf * 16, 0, lane), lds_frag(Ks, kf * 16, 1, lane)};
    frag8_t vfr[2] = {lds_frag(Vs, kf * 16, 0, lane), lds_frag(Vs, kf * 16, 1, lane)};
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) { dv[df] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[df] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int qb2 = 0; qb2 < NF / 2; ++qb2) {
      asm volatile("" ::: "memory");
      // opaque redefinition: the cross-lane reads below are loop-invariant in kf and would otherwise be hoisted
      // out of the key-fragment loop as 80 live VGPRs
#pragma unroll
      for (int k2 = 0; k2 < NR; ++k2) asm volatile("" : "+v"(lse_r[k2]), "+v"(del_r[k2]));
      int g4 = g * 4;
      asm volatile("" : "+v"(g4));
      float pt2[2][4], ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fq = 2 * qb2 + hf;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          sa = mfma16<TE>(lds_frag(Qs, fq * 16, kk, lane), kfr[kk], sa);
          pa = mfma16<TE>(lds_frag(Os, fq * 16, kk, lane), vfr[kk], pa);
        }
        float ms[4] = {1.f, 1.f, 1.f, 1.f};
        if (dp > 0.f) attn_drop4_rows(seed, bh, fq * 16 + g4, key, Tn, dp, inv_keep, ms);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = fq * 16 + g4 + j;
          const bool ok = q < Tn && key < Tn;
          const float la = __shfl(lse_r[(fq * 16) >> 6], q & 63, 64);
          const float da = __shfl(del_r[(fq * 16) >> 6], q & 63, 64);
          const float p = ok ? __expf(sa[j] * scale - la) : 0.f;
          pt2[hf][j] = p * ms[j];
          ds2[hf][j] = p * (pa[j] * ms[j] - da) * scale;
        }
      }
      const frag8_t pf = pack_frag<TE>(pt2[0], pt2[1]);
      const frag8_t sf = pack_frag<TE>(ds2[0], ds2[1]);
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        dv[df] = mfma16<TE>(lds_frag_tr(Os, troff, df, qb2), pf, dv[df]);
        dk[df] = mfma16<TE>(lds_frag_tr(Qs, troff, df, qb2), sf, dk[df]);
      }
    }
    if (key < Tn) {
      bf16_t* dstk = dqkv + ((int64_t)b * Tn + key) * gs + H + h * HD + g * 4;
      store_row4x4<TE>(dstk, dk);
      store_row4x4<TE>(dstk + H, dv);
    }
  }
}

// ===================================================================================== tiled kernels: any T
// Sequences longer than one workgroup's LDS images (T > 160: the paired-input model at T = 2*149+3 = 301, the 5 s
// clips of the large model at T = 249, evaluation utterances of up to ~145 s = 7249 frames) stream the OTHER
// sequence axis through LDS in tiles of 64 rows (double-buffered, next tile's global loads in flight under the
// current tile's MFMAs): the forward is an online-softmax (flash) loop over key tiles, the backward two kernels that
// recompute the probabilities from the saved log-sum-exp -- dQ (+ delta) over key tiles, dK / dV over query tiles --
// with the same per-fragment arithmetic, dropout stream and k-slot mapping as the single-workgroup kernels above.
// No [B, h, T, T] tensor exists anywhere; work per launch is O(T^2 d), LDS is 32-33 KiB whatever T is.
constexpr int AT_TILE = 64;

struct TileRegs { uint4 v[2]; };
// rows row0 .. row0+63 (x 64 elements) of a [*, 64]-column slab with row stride gs -> registers (2 x 16 B per thread)
__device__ __forceinline__ void tile_load(TileRegs& r, const bf16_t* g, int64_t gs, int row0, int n_valid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = row0 + (c >> 3), ch = c & 7;
    r.v[i] = make_uint4(0, 0, 0, 0);
    if (row < n_valid) r.v[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * gs + ch * 8);
  }
}
__device__ __forceinline__ void tile_store(const TileRegs& r, bf16_t* lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c >> 3, ch = c & 7;
    *reinterpret_cast<uint4*>(lds + row * 64 + ((ch ^ aswz(row)) << 3)) = r.v[i];
  }
}

template <typename TE>
__global__ __launch_bounds__(256) void attn_fwd_tiled_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                             float* __restrict__ lse, int Tn, int heads, float scale,
                                                             float dp, float inv_keep, uint64_t seed) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][AT_TILE * 64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int64_t bh = (int64_t)b * heads + h;
  frag8_t qf[2];
  reg_frag(qf, qb, gs, q, Tn, lane);
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rk, rv;
  tile_load(rk, qb + H, gs, 0, Tn);
  tile_load(rv, qb + 2 * H, gs, 0, Tn);
  tile_store(rk, Ks[0]);
  tile_store(rv, Vs[0]);
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {                              // next tile's loads fly under this tile's arithmetic
      tile_load(rk, qb + H, gs, (t + 1) * AT_TILE, Tn);
      tile_load(rv, qb + 2 * H, gs, (t + 1) * AT_TILE, Tn);
    }
    float s[4][4];
    float tmax = -INFINITY;
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc = mfma16<TE>(lds_frag(Ks[cur], fj * 16, kk, lane), qf[kk], acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = t * AT_TILE + fj * 16 + g * 4 + j;
        s[fj][j] = key < Tn ? acc[j] * scale : -INFINITY;
        tmax = fmaxf(tmax, s[fj][j]);
      }
    }
    tmax = quad_max(tmax);
    const float mn = fmaxf(m, tmax);                  // finite: every tile holds at least one valid key
    const float alpha = __expf(m - mn);               // first tile: exp(-inf) = 0
    m = mn;
    float psum = 0.f;
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) {
      float ms[4] = {1.f, 1.f, 1.f, 1.f};
      if (dp > 0.f) attn_drop4_keys(seed, bh, q, t * AT_TILE + fj * 16 + g * 4, Tn, dp, inv_keep, ms);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __expf(s[fj][j] - mn);
        psum += p;                                    // the normaliser sums the probabilities BEFORE dropout
        s[fj][j] = p * ms[j];
      }
    }
    l = l * alpha + quad_sum(psum);
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const frag8_t pf = pack_frag<TE>(s[2 * kb], s[2 * kb + 1]);
#pragma unroll
      for (int df = 0; df < 4; ++df) o[df] = mfma16<TE>(lds_frag_tr(Vs[cur], troff, df, kb), pf, o[df]);
    }
    if (t + 1 < ntile) {
      tile_store(rk, Ks[cur ^ 1]);                    // last read in iteration t-1, behind that iteration's barrier
      tile_store(rv, Vs[cur ^ 1]);
    }
    __syncthreads();
  }
  if (q < Tn) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] *= inv;
    store_row4x4<TE>(ctx + ((int64_t)b * Tn + q) * H + h * HD + g * 4, o);
    if (g == 0) lse[bh * Tn + q] = m + __logf(l);
  }
}

// dQ (and delta[q] = sum_d dO O) for 64 queries per workgroup, streaming key tiles
template <typename TE>
__global__ __launch_bounds__(256) void attn_bwd_dq_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ ctx,
                                                                const bf16_t* __restrict__ dctx,
                                                                const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                                int Tn, int heads, float scale, float dp,
                                                                float inv_keep, uint64_t seed) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][AT_TILE * 64];
  const int b = blockIdx.z, h = blockIdx.y;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const bf16_t* dob = dctx + (int64_t)b * Tn * H + h * HD;
  const bf16_t* ob = ctx + (int64_t)b * Tn * H + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int q = blockIdx.x * 64 + wave * 16 + (lane & 15);
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
  const float l = q < Tn ? lse[bh * Tn + q] : 0.f;
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rk, rv;
  tile_load(rk, qb + H, gs, 0, Tn);
  tile_load(rv, qb + 2 * H, gs, 0, Tn);
  tile_store(rk, Ks[0]);
  tile_store(rv, Vs[0]);
  __syncthreads();
  f32x4 o[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {
      tile_load(rk, qb + H, gs, (t + 1) * AT_TILE, Tn);
      tile_load(rv, qb + 2 * H, gs, (t + 1) * AT_TILE, Tn);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fj = 2 * kb + hf;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          sa = mfma16<TE>(lds_frag(Ks[cur], fj * 16, kk, lane), qf[kk], sa);
          pa = mfma16<TE>(lds_frag(Vs[cur], fj * 16, kk, lane), dof[kk], pa);
        }
        const int key0 = t * AT_TILE + fj * 16 + g * 4;
        float ms[4] = {1.f, 1.f, 1.f, 1.f};
        if (dp > 0.f) attn_drop4_keys(seed, bh, q, key0, Tn, dp, inv_keep, ms);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = (key0 + j < Tn && q < Tn) ? __expf(sa[j] * scale - l) : 0.f;
          ds2[hf][j] = p * (pa[j] * ms[j] - dl) * scale;
        }
      }
      const frag8_t pf = pack_frag<TE>(ds2[0], ds2[1]);
#pragma unroll
      for (int df = 0; df < 4; ++df) o[df] = mfma16<TE>(lds_frag_tr(Ks[cur], troff, df, kb), pf, o[df]);
    }
    if (t + 1 < ntile) {
      tile_store(rk, Ks[cur ^ 1]);
      tile_store(rv, Vs[cur ^ 1]);
    }
    __syncthreads();
  }
  if (q < Tn) store_row4x4<TE>(dqkv + ((int64_t)b * Tn + q) * gs + h * HD + g * 4, o);
}

// dK, dV for 64 keys per workgroup, streaming query tiles (Q, dO, LSE, delta)
template <typename TE>
__global__ __launch_bounds__(256) void attn_bwd_kv_tiled_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ dctx,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int Tn, int heads,
                                                                float scale, float dp, float inv_keep, uint64_t seed) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[2][AT_TILE * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Os[2][AT_TILE * 64];      // dO
  __shared__ __attribute__((aligned(16))) float lse_s[2][AT_TILE];
  __shared__ __attribute__((aligned(16))) float del_s[2][AT_TILE];
  const int b = blockIdx.z, h = blockIdx.y;
  const int H = heads * HD;
  const int64_t gs = 3 * (int64_t)H;
  const bf16_t* qb = qkv + (int64_t)b * Tn * gs + h * HD;
  const bf16_t* dob = dctx + (int64_t)b * Tn * H + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int key = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int64_t bh = (int64_t)b * heads + h;
  frag8_t kf[2], vf[2];
  reg_frag(kf, qb + H, gs, key, Tn, lane);
  reg_frag(vf, qb + 2 * H, gs, key, Tn, lane);
  const TrOff troff = tr_offsets(lane);
  const int ntile = (Tn + AT_TILE - 1) / AT_TILE;
  TileRegs rq, ro;
  float rl = 0.f, rd = 0.f;
  auto row_load = [&](int t) {
    if (threadIdx.x < AT_TILE) {
      const int r = t * AT_TILE + threadIdx.x;
      rl = r < Tn ? lse[bh * Tn + r] : 0.f;
      rd = r < Tn ? delta[bh * Tn + r] : 0.f;
    }
  };
  auto row_store = [&](int buf) {
    if (threadIdx.x < AT_TILE) { lse_s[buf][threadIdx.x] = rl; del_s[buf][threadIdx.x] = rd; }
  };
  tile_load(rq, qb, gs, 0, Tn);
  tile_load(ro, dob, H, 0, Tn);
  row_load(0);
  tile_store(rq, Qs[0]);
  tile_store(ro, Os[0]);
  row_store(0);
  __syncthreads();
  f32x4 dv[4], dk[4];
#pragma unroll
  for (int df = 0; df < 4; ++df) { dv[df] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[df] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntile) {
      tile_load(rq, qb, gs, (t + 1) * AT_TILE, Tn);
      tile_load(ro, dob, H, (t + 1) * AT_TILE, Tn);
      row_load(t + 1);
    }
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
      float pt2[2][4], ds2[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int fq = 2 * qb2 + hf;
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
        if (dp > 0.f) attn_drop4_rows(seed, bh, q0, key, Tn, dp, inv_keep, ms);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = (q0 + j < Tn && key < Tn) ? __expf(sa[j] * scale - la[j]) : 0.f;
          pt2[hf][j] = p * ms[j];
          ds2[hf][j] = p * (pa[j] * ms[j] - da[j]) * scale;
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
    if (t + 1 < ntile) {
      tile_store(rq, Qs[cur ^ 1]);
      tile_store(ro, Os[cur ^ 1]);
      row_store(cur ^ 1);
    }
    __syncthreads();
  }
  if (key < Tn) {
    bf16_t* dstk = dqkv + ((int64_t)b * Tn + key) * gs + H + h * HD + g * 4;
    store_row4x4<TE>(dstk, dk);
    store_row4x4<TE>(dstk + H, dv);
  }
}

// ------------------------------------------------------------------------------------- host
template <int NF> static size_t bwd3_lds() { return (size_t)(4 * NF * 16 * 64) * 2; }
template <int NF> static size_t fwd2_lds() { return (size_t)(2 * NF * 16 * 64 + 64 * (NF * 16 + 4)) * 2; }

template <typename K> static void set_lds(K kern, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static int attn_check(const char* nm, int B, int T, int heads, int d, int dtype, float drop_p) {
  W2V2_REQUIRE(B > 0 && T > 0 && heads > 0, "%s: bad shape", nm);
  W2V2_REQUIRE(d == HD, "%s: fused attention needs head dim 64 (got %d); use the unfused path", nm, d);
  W2V2_REQUIRE(dtype == W2V2_BF16 || dtype == W2V2_F16,
               "%s: fused attention needs 16-bit activations; the f32 parity mode uses the unfused path", nm);
  W2V2_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: bad dropout p", nm);
  return 0;
}

// T <= 160 (the 3 s training clips): one workgroup per (batch, head); longer sequences: the tiled kernels.
// W2V2_ATTN_TILED=1 sends every length to the tiled kernels (A/B runs and tests of their short-sequence edge cases).
static const bool g_attn_tiled = getenv("W2V2_ATTN_TILED") != nullptr;

#define ATTN_DISPATCH_SMALL(NFV, CALL)       \
  switch (NFV) {                             \
    case 2: { constexpr int NF = 2; CALL; } break;   \
    case 4: { constexpr int NF = 4; CALL; } break;   \
    case 6: { constexpr int NF = 6; CALL; } break;   \
    case 8: { constexpr int NF = 8; CALL; } break;   \
    default: { constexpr int NF = 10; CALL; } break; \
  }

template <typename TE>
static int attention_fwd_t(const void* qkv, void* ctx, float* lse, int B, int T, int heads, float scale, float drop_p,
                           uint64_t seed, void* stream) {
  const int nf = (int)cdiv(T, 32) * 2;
  const float ik = 1.0f / (1.0f - drop_p);
  hipStream_t st = as_stream(stream);
  if (nf <= 10 && !g_attn_tiled) {
    dim3 grid2(heads, B);
    ATTN_DISPATCH_SMALL(nf, {
      set_lds(attn_fwd2_kernel<TE, NF>, fwd2_lds<NF>());
      hipLaunchKernelGGL((attn_fwd2_kernel<TE, NF>), grid2, dim3(256), fwd2_lds<NF>(), st, (const bf16_t*)qkv, (bf16_t*)ctx,
                         lse, T, heads, scale, drop_p, ik, seed);
    });
  } else {
    dim3 grid((unsigned)cdiv(T, 64), heads, B);
    hipLaunchKernelGGL((attn_fwd_tiled_kernel<TE>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)ctx, lse, T,
                       heads, scale, drop_p, ik, seed);
  }
  W2V2_CHECK_LAUNCH("attention_fwd");
  return 0;
}

extern "C" int w2v2_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int heads, int d, float scale,
                                  float drop_p, uint64_t seed, int dtype, void* stream) {
  if (attn_check("attention_fwd", B, T, heads, d, dtype, drop_p)) return -1;
  W2V2_REQUIRE(qkv && ctx && lse, "attention_fwd: null pointer");
  W2V2_DISPATCH_16(dtype, "attention_fwd",
                   return attention_fwd_t<AT>(qkv, ctx, lse, B, T, heads, scale, drop_p, seed, stream););
  return 0;
}

template <typename TE>
static int attention_bwd_t(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv,
                           float* delta, int B, int T, int heads, float scale, float drop_p, uint64_t seed,
                           void* stream) {
  const int nf = (int)cdiv(T, 32) * 2;
  const float ik = 1.0f / (1.0f - drop_p);
  hipStream_t st = as_stream(stream);
  if (nf <= 10 && !g_attn_tiled) {
    dim3 grid2(heads, B);
    ATTN_DISPATCH_SMALL(nf, {
      set_lds(attn_bwd3_kernel<TE, NF>, bwd3_lds<NF>());
      hipLaunchKernelGGL((attn_bwd3_kernel<TE, NF>), grid2, dim3(256), bwd3_lds<NF>(), st, (const bf16_t*)qkv,
                         (const bf16_t*)ctx, (const bf16_t*)dctx, lse, (bf16_t*)dqkv, delta, T, heads, scale, drop_p,
                         ik, seed);
    });
  } else {
    dim3 grid((unsigned)cdiv(T, 64), heads, B);
    hipLaunchKernelGGL((attn_bwd_dq_tiled_kernel<TE>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)ctx,
                       (const bf16_t*)dctx, lse, (bf16_t*)dqkv, delta, T, heads, scale, drop_p, ik, seed);
    hipLaunchKernelGGL((attn_bwd_kv_tiled_kernel<TE>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)dctx,
                       lse, (const float*)delta, (bf16_t*)dqkv, T, heads, scale, drop_p, ik, seed);
  }
  W2V2_CHECK_LAUNCH("attention_bwd");
  return 0;
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
