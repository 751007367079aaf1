/*
 * w2v2_hip.h -- C ABI of libw2v2hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * wav2vec2 speaker-recognition training path of nikvaessen/w2v2-speaker.
 *
 * The reference has NO native / FFI interface: its seam is a set of torch
 * nn.Module contracts whose arithmetic lives in HuggingFace transformers / torch ATen.  Every entry
 * point below therefore cites the reference (or HF) call site whose device arithmetic it replaces:
 *   ref: = /root/reference/...      HF: = transformers/models/wav2vec2/modeling_wav2vec2.py (5.15.0)
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch types.  All pointers are DEVICE pointers unless the
 *     name ends in _host.  The caller owns every buffer (inputs, outputs, saved-for-backward,
 *     workspace); the library never allocates or frees device memory and keeps no pointer past a call.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no implicit sync.
 *   - return 0 on success, negative on error; w2v2_last_error() gives the message (thread local).
 *   - dtype codes: W2V2_F32 = 0, W2V2_BF16 = 1, W2V2_F16 = 2 (IEEE half: the precision of the reference's own
 *     fp16-AMP runs, config/experiment/speaker_wav2vec2_aam.yaml:17; its backward runs under the loss scale of
 *     w2v2_grad_scaler_*).  "act dtype" is the storage type of activations; statistics, reductions, biases,
 *     LayerNorm/GroupNorm parameters and gradients are always f32.
 *   - activations are channels-last: [B, T, C] row-major ("tokens x channels").
 */
#ifndef W2V2_HIP_H
#define W2V2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2V2_F32 0
#define W2V2_BF16 1
#define W2V2_F16 2

int w2v2_version(void);
const char* w2v2_last_error(void);

/* ------------------------------------------------------------------------------------------ GEMM
 * One tiled MFMA GEMM serves every matmul-shaped op of the path (HF:520-526 q/k/v_proj, HF:544
 * out_proj, HF:565-572 FFN, HF:429-435 projection, HF:254-272 conv layers 1-6 as implicit GEMM,
 * HF:326-368 grouped pos-conv as implicit GEMM, attention score / context products HF:438-463,
 * ref: src/optim/loss/aam_softmax.py:55 F.linear) and all of their backward products.
 *
 *   C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](n,k) )        z = 0..batch-1
 *
 * Operand storage: a 2-D array [outer][inner] with `inner` contiguous and outer stride `ld`
 * (elements).  trans = 0: outer = m (or n), inner = k.   trans = 1: outer = k, inner = m (or n).
 * The outer index may be segmented (implicit im2col of a strided conv over [B, L, C] activations):
 *   addr(outer) = (outer / seg_len) * seg_stride + (outer % seg_len) * ld      (seg_len = 0: plain)
 * Batch offsets are two-level: z -> (z / batch_inner, z % batch_inner) * (stride0, stride1).
 */
typedef struct {
  const void* ptr;
  int64_t ld;
  int64_t seg_len, seg_stride;
  int64_t stride0, stride1;
  int32_t trans;
  int32_t _pad;
} w2v2_operand;

enum {
  W2V2_EPI_NONE = 0,      /* C = alpha*acc                                              */
  W2V2_EPI_BIAS = 1,      /* C = alpha*acc + bias[n]                                    */
  W2V2_EPI_BIAS_GELU = 2, /* pre = alpha*acc + bias[n]; aux = pre (if aux); C = gelu(pre) */
  W2V2_EPI_GELU_BWD = 3,  /* C = alpha*acc * gelu'(aux[m][n])                           */
  W2V2_EPI_ADD = 4,       /* C = alpha*acc + aux[m][n]   (residual / gradient join)     */
  W2V2_EPI_SCALE_RC = 5,  /* C = alpha*acc * row_scale[m] * col_scale[n]  (AAM cosine)  */
  /* FFN pair of HF:565-572 with the activation derivative taken where the pre-activation is in f32 registers:
   * the forward product stores gelu'(pre) instead of pre (same bytes), the backward product multiplies by it
   * (1 VALU op per element instead of re-evaluating erf / exp on 30 M elements per layer) */
  W2V2_EPI_BIAS_GELU_GRAD = 6, /* pre = alpha*acc + bias[n]; aux = gelu'(pre); C = gelu(pre)   */
  W2V2_EPI_MUL = 7             /* C = alpha*acc * aux[m][n]                                    */
};

typedef struct {
  int32_t M, N, K;
  int32_t batch, batch_inner; /* batch_inner >= 1 */
  int32_t dtype_ab;           /* dtype of A and B                                        */
  int32_t dtype_c;            /* dtype of C and aux                                      */
  int32_t epilogue;
  w2v2_operand A, B;
  void* C;
  int64_t ldc, c_stride0, c_stride1;
  void* aux;
  int64_t ldaux, aux_stride0, aux_stride1;
  const float* bias;       /* [N] f32 (EPI_BIAS*)                                         */
  int64_t bias_stride1;    /* bias offset per (z % batch_inner)                           */
  const float* row_scale;  /* [M] f32, const float* col_scale [N] (EPI_SCALE_RC)          */
  const float* col_scale;
  float alpha;
  int32_t split_k;         /* >1: K split over grid.y, partial sums atomically ADDED into an
                              f32 C the caller zeroed (EPI_NONE only)                     */
  int32_t accumulate;      /* 1: C += result (f32 C, atomics; implied by split_k>1)       */
  int32_t _pad;
  /* Two-term weights (fp16 activations; 0 = off): B = B_hi + B_lo with B_lo = fp16(W - fp16(W)) stored
   * b_lo_offset ELEMENTS behind B_hi (same layout).  Output columns n >= n_ext_from run k_ext (= K) extra
   * K steps  sum_k A(m,k) * B_lo(n,k), i.e. C = A (B_hi + B_lo)^T with the weight rounding error of ~2^-22
   * instead of 2^-11.  Used for the value and output projections of the attention block, whose weight
   * rounding is the largest term of the embedding error budget (DESIGN.md "precision").  Runs on the 256x128
   * LDS-DMA ring kernel: plain NT product, K % 64 == 0, 16-byte aligned rows; anything else is an error. */
  int32_t k_ext, n_ext_from;
  int64_t b_lo_offset;
} w2v2_gemm_desc;

int w2v2_gemm(const w2v2_gemm_desc* d, void* stream);
/* Measurement hook (bench.py `roofline`): the same launch, timed by the dispatch's own begin / end timestamps
 * (hipExtLaunchKernelGGL with a start / stop event owned by the library, slot = index into its event pool) -- what
 * rocprofv3 reports for the kernel, without the ~3 us of dispatch time that two hipEventRecord calls around a launch
 * include.  Only products that run on the 256x128 ring or the phased 256x256 kernel (error otherwise).
 * w2v2_timer_read waits for slots [first_slot, first_slot + n) and writes their durations in milliseconds. */
int w2v2_gemm_timed(const w2v2_gemm_desc* d, void* stream, int slot);
int w2v2_timer_read(int first_slot, int n, float* ms_out);
/* The kernel family w2v2_gemm sends this descriptor to, from a dry run of the dispatch itself (nothing is launched):
 * 4 = phased 256x256, 2 = 256x128 ring, 1 = 128-row LDS-DMA, 3 = register-staged, 9 = exact f32 (register-staged,
 * csrc/gemm_f32.hip), 10 = exact f32 with LDS-DMA staging (csrc/gemm_f32_dma.hip); < 0 = rejected. */
int w2v2_gemm_kernel_of(const w2v2_gemm_desc* d);
/* Tuning hook (tools/gemm_shapes.py, not used by the training path): force the tile family of plain K-contiguous
 * 16-bit products -- 0 = the library's own dispatch, 1 = 128x128, 2 = 256x128 ring, 3 = 256x256x32 ring,
 * 4 = 256x256x64 phased.  Returns the previous setting. */
int w2v2_tune_gemm_kernel(int family);
/* Tools only (tools/gemm_attrib.py): time-attribution / placement variants of the 256x256x64 phased kernel for fp16
 * products -- bit 0 no DMA in the main loop, 1 no fragment reads, 2 no MFMAs, 3 no epilogue, bits 4-5 = DMA pieces of a
 * phase issued between its MFMAs, bit 6 / 7 plain / write-through epilogue stores.  Variants with bits 0-3 compute
 * garbage by design.  Returns the previous setting; 0 = the product kernel. */
int w2v2_tune_gemm_debug(int bits);
/* Tools only: force the kernel / block tile of the exact-f32 MFMA GEMM -- 0 = the library's choice; 1..4 = the
 * register-staged kernel (csrc/gemm_f32.hip) on 128x128 / 64x128 / 128x64 / 64x64; 11..15 = the LDS-DMA kernel
 * (csrc/gemm_f32_dma.hip) on (32 fi) x 128 tiles, fi = tile - 10; + 100 = the same with XCD-contiguous tile order.  A product the LDS-DMA kernel cannot take (segmented or
 * unaligned operands, K % 4 != 0, M or N <= 64) ignores codes >= 11.  Returns the previous setting. */
int w2v2_tune_gemm_f32_tile(int tile);
/* Tools only: time-attribution variants of the 256x128 ring GEMM's K loop (results are garbage) -- bit 0 no LDS-DMA pieces in
 * the steady-state loop, 1 no barrier, 2 no vmcnt wait, 3 no fragment reads.  Returns the previous setting; 0 = the product
 * kernel. */
int w2v2_tune_gemm_ring_debug(int bits);
/* Tools / tests: which kernel the LAST exact-f32 product ran on -- 0 = register-staged, else 10 fi + stages of the
 * LDS-DMA kernel (e.g. 52 = 160 x 128 tiles, two-stage ring). */
int w2v2_gemm_f32_last_kernel(void);

/* Grouped weight-gradient GEMM (the backward of HF:520-526,544,565-572 nn.Linear weights/biases):
 *   dW_p[o][i] = sum_t dY_p[t][o] * X_p[t][i]      dbias_p[o] = sum_t dY_p[t][o]   (dbias may be NULL)
 * for 1..32 (dY, X) pairs in ONE launch, 16-bit operands [tokens][features] (K-major), f32 results
 * WRITTEN (not accumulated), no split-K, no atomics -> bitwise reproducible.
 * Contract: rows [tokens, tokens_padded) of every dY / X are readable and zero
 * (tokens_padded = tokens rounded up to 64); n_out, n_in multiples of 8; 16-byte aligned rows. */
typedef struct {
  const void* dY; int64_t ld_dy;
  const void* X;  int64_t ld_x;
  float* dW;      int64_t ld_dw;
  float* dbias;
  int32_t n_out, n_in;
} w2v2_wgrad_problem;
int w2v2_wgrad_grouped(const w2v2_wgrad_problem* problems, int n, int tokens, int tokens_padded,
                       int dtype /* W2V2_BF16 or W2V2_F16: element type of dY and X */, void* stream);
/* Tools only (tools/gemm_shapes.py, tests): force the kernel of the grouped weight gradients -- 0 = the library's own
 * choice, 1 = 128x128 two-stage, 2 = 256x128 ring, 3 = 256x256x32 ring, 4 = 256x256x64 phased (5 / 6: with one / both DMA
 * pieces of a phase issued between its MFMAs).  Returns the previous value. */
int w2v2_tune_wgrad_kernel(int family);

/* ------------------------------------------------------------------------ conv feature extractor
 * Layer 0 of HF:382-419: Conv1d(1->C,k,stride,no bias) + GroupNorm(C groups == per-(b,c) over time,
 * biased var, eps) + GELU(erf)  (HF:302-323).  wav [B,N] f32 -> y [B,L,C] act dtype, channels-last.
 * Two passes, conv recomputed in both so the [B,L,C] pre-norm tensor never touches HBM:
 *   stats: per-block partial {sum, sumsq} -> partial[B][nchunk][C][2] (f32 workspace of
 *          B * w2v2_conv0_workspace_floats() floats), folded in a fixed order (f64) into
 *          mean_rstd[B][C][2] = {mean, 1/sqrt(var+eps)}: deterministic and batch-invariant.
 *   apply: normalise + GELU + store. */
int w2v2_conv0_workspace_floats(int N, int C, int k, int stride);
int w2v2_conv0_stats(const float* wav, const float* w /*[C][k]*/, float* partial, float* mean_rstd,
                     int B, int N, int C, int k, int stride, float eps, void* stream);
/* Statistics for the matrix-core convolution that w2v2_conv0_apply uses for 16-bit outputs (x and w split into
 * bf16 hi+lo pairs, xh.wh + xh.wl + xl.wh in one K=32 MFMA, relative error ~2^-16); same arguments and
 * workspace as w2v2_conv0_stats, which it calls for shapes outside C % 128 == 0, 3k <= 32.
 * k == 10 (the wav2vec2 layer): NO convolution is run -- the conv is linear, so sum(y) and sum(y^2) over time are
 * w.S and w^T R w with the 10 + 55 window moments S[j] = sum_t x[t*stride+j], R[j][j'] = sum_t x[..+j] x[..+j'] of the
 * waveform (f64, fixed order: deterministic and batch-invariant); 16 us instead of 100 at B = 66, 3 s.
 * W2V2_CONV0_NO_GRAM=1 keeps the convolution-based statistics (A/B). */
int w2v2_conv0_stats_mfma(const float* wav, const float* w, float* partial, float* mean_rstd,
                          int B, int N, int C, int k, int stride, float eps, void* stream);
int w2v2_conv0_apply(const float* wav, const float* w, const float* mean_rstd, const float* gamma,
                     const float* beta, void* y, int dtype, int B, int N, int C, int k, int stride,
                     void* stream);
/* Layer 0 of the feat_extract_norm="layer" family (HF:275-299; "-lv60" / xlsr checkpoints):
 * out[b, l, :] = GELU(LayerNorm_C(conv1d(wav, w [C][k], stride)[b, l, :] + bias)) -- the normalisation runs over the channels
 * of one frame (eps = nn.LayerNorm's default 1e-5 at the reference).  bias may be NULL (conv_bias=False).  f32 arithmetic,
 * `dtype` output; C % 8 == 0, C <= 512, k <= 16.  Forward only. */
int w2v2_conv0_layernorm_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta,
                              void* out, int B, int N, int C, int k, int stride, float eps, int dtype, void* stream);
/* Backward of layer 0 (unfrozen feature extractor): from dz = dL/d(layer-0 output) [B,L,C] compute dw [C][k],
 * dgamma [C], dbeta [C] (f32 atomics, caller zeroes); conv / GroupNorm are recomputed from the waveform and the
 * forward's mean_rstd.  sums [B][C][2] is f32 scratch. */
int w2v2_conv0_bwd(const float* wav, const float* w, const float* mean_rstd, const float* gamma,
                   const float* beta, const void* dz, float* sums, float* dw, float* dgamma, float* dbeta,
                   int dtype, int B, int N, int C, int k, int stride, void* stream);
/* col2im of a strided Conv1d data gradient: col [B*Lout][k*Cin] (= dpre @ packed weight) -> dx [B][Lin][Cin]. */
int w2v2_col2im(const void* col, void* dx, int B, int Lin, int Lout, int Cin, int k, int stride, int dtype,
                void* stream);
/* packed weight gradient [Cout][k][Cin] (f32) ADDED into the HF-layout gradient [Cout][Cin][k]. */
int w2v2_unpack_conv_grad(const float* gp, float* g, int Cout, int Cin, int k, void* stream);
/* HF conv weight [Cout][Cin][k] (f32) -> implicit-GEMM operand [Cout][k][Cin] (dtype). */
int w2v2_pack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int k, void* stream);

/* ----------------------------------------------------------------------------- LayerNorm family
 * y = LN(x + dropout(r)) * gamma + beta   (HF:429 feature_projection.layer_norm, HF:691-692 encoder
 * prologue, HF:596-601 post-LN residual blocks).  r may be NULL.  When r != NULL the pre-norm sum
 * s = x + dropout(r) is written back INTO r (saved for backward, no extra buffer).
 * Dropout (HF nn.Dropout) keeps element i when rand(seed, i) >= p and scales by 1/(1-p). */
int w2v2_layernorm_fwd(const void* x, void* r_inout, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int M, int H, float eps, float drop_p,
                       uint64_t seed, int dtype, void* stream);
/* y = GELU(LN(x) * gamma + beta), x and y [M, H] (y may alias x): the convolution layers of the feat_extract_norm="layer"
 * checkpoints (HF:275-299 Wav2Vec2LayerNormConvLayer: conv -> LayerNorm over the channels -> GELU).  Forward only: no
 * statistics are saved (the feature extractor of this family runs frozen, the reference's default). */
int w2v2_layernorm_gelu_fwd(const void* x, const float* gamma, const float* beta, void* y, int M, int H, float eps,
                            int dtype, void* stream);
/* s = pre-norm input (x if r was NULL).  Outputs: ds (grad wrt s; may alias dy), d_r = ds*dropmask
 * (optional; a plain copy of ds when drop_p == 0), dgamma/dbeta ADDED to (f32, caller zeroes): with a
 * workspace of w2v2_layernorm_bwd_workspace_floats(H) floats the column sums are folded in a fixed
 * order by a second tiny kernel (deterministic, no atomics); workspace NULL falls back to f32 atomics. */
int w2v2_layernorm_bwd_workspace_floats(int H);
int w2v2_layernorm_bwd(const void* dy, const void* s, const float* mean, const float* rstd,
                       const float* gamma, void* ds, void* d_r, float* dgamma, float* dbeta,
                       float* workspace, int M, int H, float drop_p, uint64_t seed, int dtype,
                       void* stream);
/* Deferred fold: call w2v2_layernorm_bwd with dgamma = dbeta = NULL and a workspace -> the per-block column partials
 * stay in the workspace (one workspace per LayerNorm); this entry folds up to 8 of them (same M, H) into their
 * dgamma / dbeta (+=, fixed order) in ONE launch. */
typedef struct {
  const float* partial;    /* the workspace given to w2v2_layernorm_bwd */
  float* dgamma;
  float* dbeta;
} w2v2_ln_fold;
int w2v2_layernorm_bwd_fold(const w2v2_ln_fold* entries, int n, int M, int H, void* stream);

/* ---------------------------------------------------------------------------------- elementwise */
/* nn.Dropout fwd == bwd (same mask): y = x * keep(seed,i)/(1-p); in place allowed. */
int w2v2_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* dx = dy * gelu'(pre)  (exact erf GELU, ACT2FN["gelu"]). */
int w2v2_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream);
/* dx[m][n] = dy * gelu'(pre) over a dense [M][N] matrix and colsum[n] += sum_m dx[m][n] (f32 atomics, caller zeroes):
 * GELU backward of the positional convolution fused with its bias gradient (HF:360-368).  N % 8 == 0, 16-byte aligned. */
int w2v2_gelu_bwd_colsum(const void* dy, const void* pre, void* dx, float* colsum, int M, int N, int dtype, void* stream);
/* y = x + a (gradient joins). */
int w2v2_add(const void* x, const void* a, void* y, int64_t n, int dtype, void* stream);
/* out[n] += sum_m x[m][n]   (bias gradients; f32 atomics, caller zeroes). */
int w2v2_colsum(const void* x, int64_t ld, float* out, int M, int N, int dtype, void* stream);
/* base[off_r .. off_r + n_r) = 0 for the (off, n) pairs of the int64 device table: gradient zeroing of the tensors the
 * backward ACCUMULATES into (LayerNorm gamma / beta, pos-conv bias, masked_spec_embed, LayerDrop-skipped layers) --
 * the large gradients are written, not accumulated (w2v2_wgrad_grouped), so the reference's optimizer.zero_grad()
 * (PL, torch.optim.Optimizer.zero_grad) does not have to touch them. */
int w2v2_zero_ranges(float* base, const int64_t* table, int n_ranges, int blocks_per_range, void* stream);
/* out[0] = mean of x[0..n): the scalar loss (ref: F.cross_entropy reduction="mean", aam_softmax.py:72). */
int w2v2_mean(const float* x, float* out, int n, void* stream);
/* f32 -> act dtype cast (bf16 weight copies), and strided 2-D copy/convert. */
int w2v2_cast(const float* x, void* y, int64_t n, int dtype, void* stream);
/* dst_i[C][R] = transpose(src_i[R][C]) for n matrices of one arena; table (device) holds per matrix
 * {src element offset, dst element offset, R, C}.  Refreshes the pre-transposed bf16 weight copies the
 * data-gradient GEMMs read (so dX = dY W runs K-contiguous like the forward products). */
int w2v2_transpose_many(const void* src, void* dst, const int64_t* table, int n, int blocks_per_matrix,
                        int dtype, void* stream);
/* SpecAugment time mask HF:1290-1292: h[m][:] = embed where mask[m].  Backward: d_embed += sum of
 * masked rows of dh (atomics), masked rows of dh zeroed in place. */
int w2v2_mask_fill(void* h, const uint8_t* mask, const float* embed, int M, int H, int dtype,
                   void* stream);
/* SpecAugment feature mask HF:1294-1304 (`mask_feature_prob`, ref: config/network/wav2vec2_fc.yaml:56): h[b][t][c] = 0
 * where mask[b][c] (mask [B][H] bytes, 8-byte aligned, drawn on the host by the HF sampler AFTER the time mask).  The
 * backward is the same call on the gradient. */
int w2v2_mask_feature(void* h, const uint8_t* mask, int B, int T, int H, int dtype, void* stream);
int w2v2_mask_fill_bwd(void* dh, const uint8_t* mask, float* d_embed, int M, int H, int dtype,
                       void* stream);
/* CLS token (ref: src/models/wav2vec2.py:128-140): y[b][0][:] = c, y[b][1+t][:] = x[b][t][:]. */
int w2v2_prepend_token(const void* x, void* y, float c, int B, int T, int H, int dtype, void* stream);

/* ------------------------------------------------------------------------------------- pos-conv
 * HF:326-379 grouped weight-normed Conv1d(H,H,K,pad=K/2,groups=G) as an implicit GEMM per group:
 * regroup pads and splits channels  x [B,T,H] -> xg [B,G,T+K-1,H/G] (zero pad `pad_left` frames left).
 * weightnorm_pack: w = g*v/||v||_(per tap) -> fwd operand wf[G][co][tap][ci] and the flipped
 * operand wb[G][ci][K-1-tap][co] used by the data gradient; also returns inv norms.
 * weightnorm_bwd: from the packed weight gradient dwf[G][tap][ci][co] (f32; = the GEMM
 *   dwf_g[(tap,ci)][co] = sum_t xg_g[t][(tap,ci)] * dy_g[t][co], large dimension as M) -> dg[K],
 *   dv[H][H/G][K] (written, not added).
 * `sumsq` / `dot` are f32 scratch of w2v2_weightnorm_scratch_floats(H, G, K) = (1 + H)*K floats: [0,K) the per-tap
 * result, the rest per-block partials (one block per output channel) that are folded in a fixed order (bitwise
 * reproducible packed weights). */
int w2v2_posconv_regroup(const void* x, void* xg, int B, int T, int H, int G, int K, int pad_left,
                         int dtype, void* stream);
/* Weight gradient of the grouped positional conv as a correlation on the matrix cores (replaces the implicit-GEMM
 * call for 16-bit activations): dY [B*T, H] = gradient at the conv output (before the weight-norm), xg [B, G, T+K-1, H/G]
 * = posconv_regroup(x, pad_left = K/2); dwf [G][K*Cg][Cg] f32 is OVERWRITTEN (row (tap, ci), column co).
 * Cg = H/G in {16, 32, 48, 64}, K a multiple of 16.  (ref: the autograd of HF:360-368 `self.conv`.) */
int w2v2_posconv_wgrad(const void* dY, const void* xg, float* dwf, int B, int T, int H, int G, int K,
                       int dtype /* W2V2_BF16 or W2V2_F16 */, void* stream);
/* The grouped positional convolution itself (HF:326-379) and its data gradient as a DIRECT convolution with the padded
 * image of one (utterance, group) resident in LDS (csrc/posconv_direct.hip; w2v2-base geometry: Cg = 48, K = 128):
 *   out[b, t, g*Cg + co] = epi( sum_{tap, ci} xg[b, g, t + tap, ci] * w[g][co][tap*Cg + ci] )
 * xg = posconv_regroup(x, pad_left) [B][G][T+K-1][Cg]; w = the packed (weight-normed) weights [G][Cg][K*Cg] of
 * w2v2_weightnorm_pack (forward: wf; data gradient: wb over the regrouped dY).  mode 0: epi = GELU(. + bias[g*Cg+co]),
 * aux (may be NULL) receives the pre-activation; mode 1: epi = . + aux (aux may alias out).  16-bit activations;
 * bit-equal to the implicit GEMM of w2v2_gemm over the same operands.  Other geometries: use the implicit GEMM.
 * Contract: ldc a multiple of 8 elements; xg, w, out and aux 16-byte aligned (else an error is returned). */
int w2v2_posconv_direct(const void* xg, const void* w, void* out, void* aux, const float* bias, int B, int T, int G,
                        int Cg, int K, int64_t ldc, int mode, int dtype, void* stream);
int64_t w2v2_weightnorm_scratch_floats(int H, int G, int K);
int w2v2_weightnorm_pack(const float* g, const float* v, float* sumsq /*[(1+H)*K]*/, void* wf,
                         void* wb, int H, int G, int K, int dtype, void* stream);
int w2v2_weightnorm_bwd(const float* g, const float* v, const float* sumsq, const float* dwf,
                        float* dot /*[(1+H)*K]*/, float* dg, float* dv, int H, int G, int K,
                        void* stream);

/* ------------------------------------------------------------------------------------ attention
 * Unfused path (scores/context products go through w2v2_gemm): row softmax with dropout on the
 * probabilities (HF:455-457).  s [rows][T] f32 scores (already scaled by d^-1/2 via alpha),
 * p = softmax(s) (act dtype), p_drop = dropout(p) (only if drop_p > 0).
 * bwd: dS = P * (dP - sum(dP*P)) with dP = dPdrop*mask/(1-p); result f32 -> act dtype ds. */
int w2v2_softmax_fwd(const float* s, void* p, void* p_drop, int64_t rows, int T, int64_t ld,
                     float drop_p, uint64_t seed, int dtype, void* stream);
int w2v2_softmax_bwd(const float* dp_drop, const void* p, void* ds, int64_t rows, int T, int64_t ld,
                     float drop_p, uint64_t seed, int dtype, void* stream);
/* Fused multi-head self-attention over QKV [B,T,3,heads,d] (HF:438-548 minus the projections),
 * one workgroup per (batch, head, query tile); saves row log-sum-exp for the backward.
 * ctx [B,T,heads*d].  bwd writes dqkv [B,T,3,heads,d]. */
int w2v2_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int heads, int d,
                       float scale, float drop_p, uint64_t seed, int dtype, void* stream);
int w2v2_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                       void* dqkv, float* delta /*[B,heads,T] scratch*/, int B, int T, int heads,
                       int d, float scale, float drop_p, uint64_t seed, int dtype, void* stream);

/* -------------------------------------------------------------------------------------- pooling
 * ref: src/layers/pooling.py:24-44,74-80,118-136.  x [B,T,H] act dtype -> out f32.
 * mode 0: mean+std  -> [B,2H] = cat(std_unbiased, mean)  (std FIRST, quirk Q1)
 * mode 1: mean      -> [B,H]       mode 2: max -> [B,H]
 * mode 3: first     -> [B,H]       mode 4: last / "middle" (quirk Q2) -> [B,H]
 * mode 5: quantile  -> [B,5H] = the 0 / .25 / .5 / .75 / 1 quantiles over time, quantile-major
 *         (ref: src/layers/pooling.py:51-67, torch.quantile with linear interpolation)
 * mode 16 + t: frame t -> [B,H]  (IndexPool1D "random", ref: src/layers/pooling.py:125-126,150-154: the HOST draws t
 *         with random.randint like the reference; NoPooling, :160-166, is the same call on the [B*T, 1, H] view) */
int w2v2_pool_fwd(const void* x, float* out, int B, int T, int H, int mode, int dtype, void* stream);
/* dx [B,T,H] act dtype from dout f32 and the forward output (std/mean reused; max needs x). */
int w2v2_pool_bwd(const void* x, const float* out, const float* dout, void* dx, int B, int T, int H,
                  int mode, int dtype, void* stream);

/* ------------------------------------------------------------------- attentive statistics pooling
 * ref call site: src/layers/pooling.py:87-106 (AttentiveStatPool1D) -> speechbrain 0.5.x
 * AttentiveStatisticsPooling(C, attention_channels A = 128, global_context=True); speechbrain is not part of the
 * reference tree: the arithmetic follows its published definition (oracle.attentive_stat_pool).
 *   ctx [B][2C] = {mean_t x, std_t x} (biased, clamp 1e-12);  a = Wx x_t + (W1[:, C:] ctx + b1) (GEMM + cb);
 *   h = tanh(BatchNorm(relu(a))) with batch statistics over all B*T rows;  s = W2 h + b2 (GEMM);
 *   w = softmax_t s;  out [B][2C] = {sum_t w x, sqrt(clamp(sum_t w (x - mean)^2, 1e-12))}  (mean first).
 * x, a, h, s and their gradients are [B*T][.] act dtype; parameters / statistics f32.  The Wx / W2 products and
 * their data / weight gradients are w2v2_gemm / w2v2_wgrad_grouped calls made by the host. */
int w2v2_asp_context(const void* x, float* ctx, int B, int T, int C, int dtype, void* stream);
int w2v2_asp_context_bias(const float* ctx, const float* w1 /*[A][3C]*/, const float* b1, float* cb /*[B][A]*/,
                          int B, int A, int C, void* stream);
int w2v2_asp_bn_workspace_floats(int M, int A);
/* batch statistics of relu(a_pre) -> mean_rstd[A][2]; running = {mean[A], var[A]} updated like torch
 * BatchNorm1d (momentum, unbiased variance) or NULL */
int w2v2_asp_bn_stats(const void* a_pre, float* workspace, float* mean_rstd, float* running, int M, int A,
                      float eps, float momentum, int dtype, void* stream);
int w2v2_asp_bn_eval_stats(const float* running, float* mean_rstd, int A, float eps, void* stream);
int w2v2_asp_bn_tanh(const void* a_pre, const float* mean_rstd, const float* gamma, const float* beta, void* h,
                     int M, int A, int dtype, void* stream);
/* dh -> da_pre through tanh, BatchNorm (batch statistics) and relu; writes dgamma[A], dbeta[A].  tanh is
 * recomputed from a_pre (1 - h^2 from a stored bf16 h has no precision for saturated units). */
int w2v2_asp_bn_bwd(const void* dh, const void* a_pre, const float* mean_rstd, const float* gamma,
                    const float* beta, float* workspace, float* dgamma, float* dbeta, void* da, int M, int A,
                    int dtype, void* stream);
/* stats [B][C][2] = {max_t s, sum_t exp(s - max)} saved for the backward */
int w2v2_asp_pool_fwd(const void* x, const void* s, float* out, float* stats, int B, int T, int C, int dtype,
                      void* stream);
/* dout [B][2C] -> ds (through the softmax) and the direct part of dx (both written) */
int w2v2_asp_pool_bwd(const void* x, const void* s, const float* out, const float* stats, const float* dout,
                      void* ds, void* dx, int B, int T, int C, int dtype, void* stream);
/* context path: dW1[:, C:] (written), dx += d(mean, std) terms; scratch = B*A + B*2C floats */
int w2v2_asp_context_bwd(const void* x, const float* ctx, const void* da, const float* w1, float* dw1, void* dx,
                         float* scratch, int B, int T, int C, int A, int dtype, void* stream);

/* Paired-input head (ref: src/lightning_modules/speaker/wav2vec2_paired_input.py:200-206 nn.Linear(H,1) on the CLS
 * token + src/optim/loss/binary_cross_entropy.py:24-40): prob = sigmoid(emb.w + b), loss_rows = BCE-with-logits per
 * pair (mean taken by the caller); gradient outputs (all or none): dlogit[B] = (p-y)/B, demb[B][H], dw[H], db[1],
 * all multiplied by *loss_scale when that device pointer is given. */
int w2v2_bce_head_fwd_bwd(const float* emb, const float* w, const float* b, const int64_t* label, float* prob,
                          float* loss_rows, float* dlogit, float* demb, float* dw, float* db, int B, int H,
                          const float* loss_scale, void* stream);

/* ------------------------------------------------------------------------------ ECAPA-TDNN pieces
 * ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85 -> speechbrain 0.5.x ECAPA_TDNN (not part of the reference
 * tree; restated in oracle/ecapa_oracle.py).  Channels-last [B*T][C] activations; `ld*` = row stride in elements so
 * Res2Net channel slices and the MFA concatenation are views of their parent tensors.
 * BatchNorm1d over the M rows of a channels-last [M][C] view (C, strides multiples of 8), optionally on relu(a)
 * (TDNNBlock = conv -> ReLU -> BatchNorm).  train = 1: batch statistics (biased variance) through `workspace`
 * partial sums, mean_rstd[C][2] written for the backward, running = {mean[C], var[C]} updated like torch or NULL;
 * train = 0: the running statistics (BatchNorm1d.eval()).  Two launches (partial sums, fold + apply), deterministic. */
int w2v2_bn_workspace_floats(int M, int C);
int w2v2_bn_fwd(const void* a, int64_t lda, float* workspace, float* mean_rstd, float* running, const float* gamma,
                const float* beta, void* y, int64_t ldy, int M, int C, float eps, float momentum, int relu, int train,
                int dtype, void* stream);
/* dy -> da (through BatchNorm and the optional relu); writes dgamma[C], dbeta[C].  colsum_partial (may be NULL):
 * [w2v2_bn_colsum_rows(M, C)][C] f32, WRITTEN with the column sums of da over each row block -- the bias gradient of the
 * convolution in front of the BatchNorm (TDNNBlock = conv -> ReLU -> BatchNorm) is their sum: a w2v2_colsum over
 * M / 256 rows (M / 64 for the narrow Res2Net slices, whose row blocks are shorter so that they fill the chip) instead of
 * a second pass over the M rows of da. */
int w2v2_bn_colsum_rows(int M, int C);
int w2v2_bn_bwd(const void* dy, int64_t lddy, const void* a, int64_t lda, const float* mean_rstd, const float* gamma,
                float* workspace, float* dgamma, float* dbeta, void* da, int64_t ldda, int M, int C, int relu,
                float* colsum_partial, int dtype, void* stream);
/* The same with the output gradient given as dy + dy2 (same shape, own row strides; both kernels add them as they read):
 * a Res2Net chunk's output feeds the next chunk and the block's concatenation, so its gradient has two sources. */
int w2v2_bn_bwd_sum(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* a, int64_t lda,
                    const float* mean_rstd, const float* gamma, float* workspace, float* dgamma, float* dbeta, void* da,
                    int64_t ldda, int M, int C, int relu, float* colsum_partial, int dtype, void* stream);
/* Conv1d(padding="same", padding_mode="reflect", dilation d, odd k) as im2col + GEMM:
 * col[(b,t)][j*Cin + c] = x[b][reflect(t + (j - (k-1)/2) d)][c]; col2im is its exact adjoint (gather, deterministic) */
int w2v2_im2col_reflect(const void* x, int64_t ldx, void* col, int B, int T, int Cin, int k, int dilation, int dtype,
                        void* stream);
/* The same over x + x2 (two row-strided operands of one shape, summed tap by tap): the input x_i + y_{i-1} of a Res2Net
 * chunk (speechbrain Res2NetBlock.forward) without a pass that materialises the sum. */
int w2v2_im2col_reflect_sum(const void* x, int64_t ldx, const void* x2, int64_t ldx2, void* col, int B, int T, int Cin,
                            int k, int dilation, int dtype, void* stream);
int w2v2_col2im_reflect(const void* dcol, void* dx, int64_t lddx, int B, int T, int Cin, int k, int dilation,
                        int accumulate, int dtype, void* stream);
/* y[m][0..C) = a[m][0..C) + b[m][0..C) over row-strided views (Res2Net cumulative adds); b == NULL: y = a (a strided
 * copy: the pass-through chunk of a Res2Net block, the SE input slice). */
int w2v2_add_strided(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int M, int C,
                     int dtype, void* stream);
/* squeeze-excitation gate g [B][C] f32: y = x * g;  dg = sum_t dout * x;  dx = dout * g + ds / T */
int w2v2_se_scale(const void* x, const float* g, void* y, int B, int T, int C, int dtype, void* stream);
int w2v2_se_bwd_gate(const void* dout, const void* x, float* dg, int B, int T, int C, int dtype, void* stream);
int w2v2_se_bwd_x(const void* dout, const float* g, const float* ds, void* dx, int B, int T, int C, int dtype,
                  void* stream);
/* f32 vectors; mode 0 relu, 1 sigmoid; the backward takes the forward output */
int w2v2_act_fwd(const float* x, float* y, int64_t n, int mode, void* stream);
int w2v2_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int mode, void* stream);

/* ---------------------------------------------------------------------------------------- skinny linear layers
 * Exact-f32 nn.Linear / Conv1d(k=1) over a handful of rows (one per utterance): the SE bottleneck and the embedding
 * layer of ECAPA (speechbrain SEBlock / ECAPA_TDNN.fc) and the hidden fc_list of the wav2vec2 head (ref:
 * src/lightning_modules/speaker/wav2vec2_fc.py:108-141).  x [B][K], W [N][K], y [B][N] row-major, K % 4 == 0.
 * act: 0 none, 1 relu, 2 sigmoid.  fwd: y = act(x W^T + bias).  The backward entries take the forward OUTPUT y and
 * use dy' = dy * act'(y):  bwd_x: dx = dy' W;  bwd_w: dW (+)= dy'^T x, dbias (+)= column sums of dy' (dbias may be
 * NULL). */
int w2v2_skinny_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int N, int K, int act,
                           void* stream);
int w2v2_skinny_linear_bwd_x(const float* dy, const float* y, const float* W, float* dx, int B, int N, int K, int act,
                             void* stream);
int w2v2_skinny_linear_bwd_w(const float* dy, const float* y, const float* x, float* dW, float* dbias, int B, int N,
                             int K, int act, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------- heads
 * Row inverse L2 norms 1/max(||x||,1e-12) (F.normalize, ref: src/optim/loss/aam_softmax.py:55). */
int w2v2_row_invnorm(const void* x, int64_t ld, float* inv, int rows, int cols, int dtype,
                     void* stream);
/* AAM margin + scale + softmax + CE (ref: aam_softmax.py:57-72) on cos [B][ldc] f32 (in: cosine).
 * margin < 0 selects the plain CE head (ref: src/optim/loss/cross_entropy.py:27-31; `cos` holds
 * logits, scale ignored).  Outputs: softmax [B][ldc] f32, loss_rows [B] f32 (mean is the caller's),
 * and for the backward, with g = dLoss/dcos (loss = mean over B):
 *   dcos_w[b][c] = g * inv_w[c]   (A operand of the embedding-gradient GEMM; inv_w NULL -> g)
 *   dcos_x[b][c] = g * inv_x[b]   (A operand of the weight-gradient GEMM;    inv_x NULL -> g)
 *   rowdot[b] = sum_c g*cos,  colprod[b][c] = g*cos ([B][C] f32, row stride C: the caller folds the rows in a fixed
 *   order -- w2v2_colsum with B <= 128 has one writer per column -- into coldot[c] = sum_b g*cos for
 *   w2v2_normalize_bwd; no atomics, so the head's weight gradient is bitwise reproducible); any may be NULL.
 * loss_scale (device pointer, may be NULL = 1): g is multiplied by *loss_scale (w2v2_grad_scaler_*); the loss
 * rows and the softmax are never scaled.  Labels outside [0, C) give loss = NaN for that row and zero gradients.
 * correct_rows [B] (may be NULL): 1 if argmax_c softmax[b][c] == label[b] else 0 -- the reference's train_acc
 * (ref: src/lightning_modules/speaker/speaker_recognition_module.py:296-307). */
int w2v2_aam_softmax_fwd_bwd(const float* cos, const int64_t* label, float* softmax,
                             float* loss_rows, void* dcos_w, void* dcos_x, const float* inv_x,
                             const float* inv_w, float* rowdot, float* colprod, int B, int C,
                             int64_t ldc, float margin, float scale, const float* loss_scale,
                             float* correct_rows, int easy_margin /* ref: aam_softmax.py:60-61 */, int dtype, void* stream);
/* F.normalize backward: dx = inv[r] * (g[r] - x[r] * inv[r] * dot[r]);  g, dx f32 (dx written or
 * added), x f32 or act dtype. */
int w2v2_normalize_bwd(const float* g, const void* x, int64_t ldx, const float* inv,
                       const float* dot, float* dx, int rows, int cols, int x_dtype, int add,
                       void* stream);
/* Class-weight gradient of the AAM head in one launch (ref: the autograd of src/optim/loss/aam_softmax.py:55 through
 * F.normalize(W)): dW[c][e] = inv_w[c] * (sum_b dcos_x[b][c] * emb[b][e] - W[c][e] * inv_w[c] * sum_b colprod[b][c]).
 * dcos_x [B][ldc] and emb [B][E] in `dtype` (the head's 16-bit operands, or f32), colprod [B][C] f32 = g * cos per element
 * (w2v2_aam_softmax_fwd_bwd), W [C][E] f32 master weights, dW [C][E] f32 WRITTEN.  E % 8 == 0, 16-byte aligned operands;
 * fixed summation order. */
int w2v2_aam_dw(const void* dcos_x, int64_t ldc, const void* emb, const float* colprod, const float* W,
                const float* inv_w, float* dW, int B, int C, int E, int dtype, void* stream);

/* ------------------------------------------------------------------------------------ optimiser
 * torch.optim.Adam (ref: config/optim/algo/adam.yaml, src/main.py:323-335) over one flat f32
 * parameter arena; also refreshes the 16-bit copy (pb_dtype = W2V2_BF16 / W2V2_F16) the GEMMs read
 * (pb may be NULL).  grad_scale folds the 1/world_size of the data-parallel average.
 * scaler_state (may be NULL): the 4-float device record of w2v2_grad_scaler_*: gradients are divided by
 * state[0] and the whole step is skipped when state[1] != 0. */
/* skip_slot (0 = off, 4 or 5): torch's GradScaler does not call optimizer.step() on an overflow, so Adam's step count
 * must not advance on skipped steps.  With skip_slot set the kernel rebuilds both bias corrections from
 * t = step - scaler_state[skip_slot] (the host's `step` counts every call; the 8-float record counts the skipped ones
 * per parameter range: [4] head, [5] body -- see w2v2_grad_scaler_update) and ignores bias_corr1 / bias_corr2. */
int w2v2_adam_step(float* p, const float* g, float* m, float* v, void* pb, int pb_dtype, int64_t n, float lr,
                   float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                   float grad_scale, const float* scaler_state, int step, int skip_slot, void* stream);

/* Dynamic loss scaling for fp16 activations = torch.cuda.amp.GradScaler, which PL `precision: 16` of the
 * reference's runs installs (ref: config/experiment/speaker_wav2vec2_aam.yaml:17).
 * state (device, 4 floats, or 8 with per-range skip counts) = {scale, found_inf, growth_tracker, skipped_steps
 * [, skipped_head, skipped_body, 0, 0]}; the heads multiply the loss
 * gradient by state[0] (w2v2_aam_softmax_fwd_bwd / w2v2_bce_head_fwd_bwd `loss_scale`), so every gradient of the
 * step carries it.  check: state[1] = 1 if any of g[0..n) is non-finite.  update (after the optimiser step):
 * found_inf ? scale *= backoff : (every growth_interval clean steps scale *= growth); clears found_inf.
 * No host synchronisation anywhere. */
/* lo[off_i .. off_i + n_i) = fp16(p[..] - fp16(p[..])) for every (off_i, n_i) of `table` ([n_ranges][2] int64, device):
 * the residual plane of the two-term weights above (p = f32 master arena, lo = a 16-bit arena of the same layout). */
int w2v2_weight_residual(const float* p, void* lo, const int64_t* table, int n_ranges, int dtype, void* stream);
int w2v2_grad_scaler_check(const float* g, int64_t n, float* state, void* stream);
/* skipped_ranges: bit 0 / bit 1 = on an overflow also count the skipped step in state[4] / state[5] (records of 8
 * floats; pass 0 for the plain 4-float record). */
int w2v2_grad_scaler_update(float* state, float growth, float backoff, int growth_interval, int skipped_ranges,
                            void* stream);

/* ------------------------------------------------------------------------------------ collective
 * ref: config/trainer/trainer.yaml:6-12 (PL `accelerator: ddp`, one process per GPU): the SUM all-reduce of the gradient
 * buckets is the only collective of a data-parallel step.  The repo's default binding issues it through
 * torch.distributed ("nccl" == RCCL); these three calls run the same collective through the C ABI alone, on RCCL over
 * xGMI (librccl.so resolved with dlopen at the first call -- no link-time dependency; W2V2_RCCL_LIB overrides the path).
 *   rank 0: w2v2_comm_unique_id(id) -> ship the 128 bytes to every rank out of band (file, socket, launcher env)
 *   every rank: w2v2_comm_init(&c, id, rank, world, device)             (collective: all ranks must call it)
 *   per bucket: w2v2_allreduce_async(c, grad + off, n, comm_stream)     (in place, f32, enqueued on the given stream;
 *               order the stream against the backward with events, like trainer.BucketAllReducer does)
 *   at start-up / after loading a checkpoint on one rank: w2v2_broadcast_async(c, buf, nbytes, root, stream) of the
 *               parameter arena, the optimiser moments and the loss-scale record (in place, any dtype: bytes) -- what
 *               PL's DDP wrapper does implicitly when it wraps the module (identical replicas)
 *   w2v2_comm_destroy(c)
 * Multi-rank operation of these entry points is covered by construction + a world-size-1 communicator on the 1-GPU test
 * boxes; the first check on an N-GPU node is tools/scale_sweep.sh (2 ranks against torch.distributed, bitwise). */
typedef struct w2v2_comm w2v2_comm;
int w2v2_comm_unique_id(void* id_host_128);
int w2v2_comm_init(w2v2_comm** comm, const void* id_host_128, int rank, int world, int device);
/* Loop-back communicator (tests / single-GPU rehearsals; no RCCL involved): stands for rank 0 of a `world`-rank job whose
 * peers hold bit-identical buffers -- w2v2_allreduce_async multiplies the buffer by `world` on the stream (the SUM of
 * `world` equal contributions), w2v2_broadcast_async leaves it as it is.  It lets the world > 1 code path of a caller
 * (bucket order, side stream, 1/world scaling, start-up broadcast) run on a box with one GPU, where RCCL refuses two
 * ranks per device. */
int w2v2_comm_init_loopback(w2v2_comm** comm, int world, int device);
int w2v2_allreduce_async(w2v2_comm* comm, float* buf, int64_t n, void* stream);
int w2v2_broadcast_async(w2v2_comm* comm, void* buf, int64_t nbytes, int root, void* stream);
int w2v2_comm_destroy(w2v2_comm* comm);
/* Tools only (tools/overlap_rehearsal.py): a stand-in for the RCCL channels of one bucket's all-reduce on a 1-GPU box --
 * `channels` workgroups holding `lds_bytes` of LDS each stream the n floats of `buf` through themselves in place (values
 * unchanged), paced to `gbps` GB/s in total (0 = unpaced).  Measures what the collective's residency costs the compute
 * stream; moves nothing between GPUs. */
int w2v2_traffic_probe(float* buf, int64_t n, int channels, int lds_bytes, float gbps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
