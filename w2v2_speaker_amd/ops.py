"""Thin tensor-level wrappers over the C ABI (include/w2v2_hip.h).

PyTorch is plumbing here: device memory, streams.  Every op enqueues hand-written HIP kernels on the
current torch stream; nothing falls back to torch / CPU arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (BF16, EPI_ADD, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD, EPI_GELU_BWD, EPI_MUL, EPI_NONE, EPI_SCALE_RC, F16, F32,
                   GemmDesc)

POOL_MODES = {"mean+std": 0, "mean": 1, "max": 2, "first": 3, "first+cls": 3, "last": 4, "middle": 4, "quantile": 5}
POOL_WIDTH = {0: 2, 5: 5}      # output features per input feature (default 1)
POOL_INDEX_BASE = 16           # mode POOL_INDEX_BASE + t selects frame t (IndexPool1D "random")


def lib():
    return _lib.load()


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def is16(dtype: torch.dtype) -> bool:
    """A 16-bit activation format (bf16 or IEEE fp16): the matrix-core kernels exist for both."""
    return dtype in (torch.bfloat16, torch.float16)


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("w2v2_speaker_amd ops need tensors on the GPU (no CPU fallback exists)")


class Gemm:
    """A prebuilt GEMM descriptor over fixed buffers: build once, launch every step."""

    def __init__(self, M: int, N: int, K: int, A: torch.Tensor, B: torch.Tensor, Cmat: torch.Tensor, *,
                 lda: int, ldb: int, ldc: int, transA: bool = False, transB: bool = False,
                 batch: int = 1, batch_inner: int = 1,
                 a_strides: Tuple[int, int] = (0, 0), b_strides: Tuple[int, int] = (0, 0),
                 c_strides: Tuple[int, int] = (0, 0),
                 a_seg: Tuple[int, int] = (0, 0), b_seg: Tuple[int, int] = (0, 0),
                 epilogue: int = EPI_NONE, bias: Optional[torch.Tensor] = None, bias_stride1: int = 0,
                 aux: Optional[torch.Tensor] = None, ldaux: int = 0, aux_strides: Tuple[int, int] = (0, 0),
                 row_scale: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None,
                 alpha: float = 1.0, split_k: int = 1, accumulate: bool = False,
                 b_lo: Optional[torch.Tensor] = None, n_ext_from: int = 0):
        """b_lo: residual plane of two-term weights (same layout as B, b_lo = T(W - T(W))): output columns
        n >= n_ext_from are computed as A (B + b_lo)^T (include/w2v2_hip.h, w2v2_gemm_desc.k_ext)."""
        _dev(A, B, Cmat, bias, aux, row_scale, col_scale, b_lo)
        if dt(A) != dt(B):
            raise TypeError("GEMM operands must share a dtype")
        if aux is not None and aux.dtype != Cmat.dtype:
            raise TypeError("aux must have the dtype of C")
        for t in (bias, row_scale, col_scale):
            if t is not None and t.dtype != torch.float32:
                raise TypeError("bias / scales are f32")
        d = GemmDesc()
        d.M, d.N, d.K = M, N, K
        d.batch, d.batch_inner = batch, max(1, batch_inner)
        d.dtype_ab, d.dtype_c, d.epilogue = dt(A), dt(Cmat), epilogue
        d.A.ptr, d.A.ld, d.A.trans = A.data_ptr(), lda, int(transA)
        d.A.seg_len, d.A.seg_stride = a_seg
        d.A.stride0, d.A.stride1 = a_strides
        d.B.ptr, d.B.ld, d.B.trans = B.data_ptr(), ldb, int(transB)
        d.B.seg_len, d.B.seg_stride = b_seg
        d.B.stride0, d.B.stride1 = b_strides
        d.C, d.ldc = Cmat.data_ptr(), ldc
        d.c_stride0, d.c_stride1 = c_strides
        d.aux, d.ldaux = _p(aux), ldaux
        d.aux_stride0, d.aux_stride1 = aux_strides
        d.bias, d.bias_stride1 = _p(bias), bias_stride1
        d.row_scale, d.col_scale = _p(row_scale), _p(col_scale)
        d.alpha, d.split_k, d.accumulate = alpha, split_k, int(accumulate)
        d.k_ext, d.n_ext_from, d.b_lo_offset = 0, 0, 0
        if b_lo is not None:
            assert b_lo.dtype == B.dtype and (b_lo.data_ptr() - B.data_ptr()) % B.element_size() == 0
            d.k_ext, d.n_ext_from = K, n_ext_from
            d.b_lo_offset = (b_lo.data_ptr() - B.data_ptr()) // B.element_size()
        self.desc = d
        self._ref = C.byref(d)
        self._keep = (A, B, Cmat, bias, aux, row_scale, col_scale, b_lo)   # keep buffers alive
        self._fn = lib().w2v2_gemm
        self.flops = 2.0 * M * N * K * batch
        # algorithmic HBM bytes: each operand read once, C written once (+ the aux tensor read or written once)
        esz = A.element_size()
        self.bytes = float(batch) * ((M * K + N * K) * esz + M * N * Cmat.element_size() * (2 if aux is not None else 1))
        # which template instantiation of csrc/gemm.hip this descriptor launches (for profiling)
        lp = dt(A) != F32
        self.kernel_class = ("mfma16" if lp else "f32") + "_" + ("t" if transA else "n") + ("n" if transB else "t")
        self.narrow = N <= 64
        self.out_is_act = is16(Cmat.dtype)
        # which kernel this descriptor launches: asked of the library's own dispatch (a dry run of w2v2_gemm, round 5 --
        # the Python mirror of its heuristics could disagree with an env-tuned or differently sized device)
        fam = lib().w2v2_gemm_kernel_of(self._ref)
        self.kernel_name = {4: "gemm16_phased_256x256_kernel", 2: "gemm16_ring_256x128_kernel", 1: "gemm16_dma_128_kernel",
                            3: "gemm16_regstage_kernel", 9: "gemm_f32_mfma_kernel",
                            10: "gemm_f32_dma_kernel"}.get(fam, "gemm16_regstage_kernel" if lp
                                                                                       else "gemm_f32_mfma_kernel")

    _prof = None
    _TIMED_KERNELS = ("gemm16_ring_256x128_kernel", "gemm16_phased_256x256_kernel")
    _log = None        # tools/gemm_instep.py: when a list, every launch appends its shape key (launch order)

    def key(self) -> dict:
        d = self.desc
        return {"kind": "gemm", "M": d.M, "N": d.N, "K": d.K, "batch": d.batch, "epi": d.epilogue,
                "two_term": int(d.k_ext != 0), "n_ext_from": d.n_ext_from, "aux": int(bool(d.aux)),
                "kernel": self.kernel_name, "flops": self.flops + 2.0 * d.M * max(0, d.N - d.n_ext_from) * d.k_ext,
                "alg_flops": self.flops, "bytes": self.bytes}

    @classmethod
    def profile_begin(cls, select) -> None:
        """Time every launch for which select(gemm) is true: by the dispatch's own begin / end timestamps
        (w2v2_gemm_timed: the duration rocprofv3 reports) for the two kernels that have the hook, with a pair of HIP
        events on the launch stream otherwise."""
        cls._prof = {"select": select, "events": [], "slots": []}

    @classmethod
    def profile_end(cls) -> dict:
        prof, cls._prof = cls._prof, None
        if not prof or not (prof["events"] or prof["slots"]):
            return {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "by_kernel": {}}
        torch.cuda.synchronize()
        out = {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "by_kernel": {}}
        timed = []
        if prof["slots"]:
            import ctypes
            buf = (ctypes.c_float * len(prof["slots"]))()
            _lib.check(lib().w2v2_timer_read(0, len(prof["slots"]), ctypes.cast(buf, ctypes.c_void_p)), "timer_read")
            timed = [(float(buf[i]),) + prof["slots"][i] for i in range(len(prof["slots"]))]
        for ms, f, nb, name in [(a.elapsed_time(b), f, nb, name) for a, b, f, nb, name in prof["events"]] + timed:
            k = out["by_kernel"].setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for d in (out, k):
                d["launches"] += 1
                d["ms"] += ms
                d["flops"] += f
                d["bytes"] += nb
        return out

    def __call__(self) -> None:
        prof = Gemm._prof
        if Gemm._log is not None:
            Gemm._log.append(self.key())
        if (prof is not None and prof["select"](self) and self.kernel_name in Gemm._TIMED_KERNELS
                and hasattr(lib(), "w2v2_gemm_timed")):
            rc = lib().w2v2_gemm_timed(self._ref, stream(), len(prof["slots"]))
            prof["slots"].append((self.flops, self.bytes, self.kernel_name))
        elif prof is not None and prof["select"](self):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self._fn(self._ref, stream())
            e1.record()
            prof["events"].append((e0, e1, self.flops, self.bytes, self.kernel_name))
        else:
            rc = self._fn(self._ref, stream())
        if rc:
            _lib.check(rc, "gemm")


def gemm(*args, **kw) -> None:
    Gemm(*args, **kw)()


class WgradGroup:
    """Prebuilt grouped weight-gradient launch: problems = [(dY, X, dW, dbias or None), ...] with
    dY [Tpad, n_out] / X [Tpad, n_in] bf16 (rows >= tokens zero) and dW [n_out, n_in] f32."""

    def __init__(self, problems, tokens: int, tokens_padded: int):
        n = len(problems)
        arr = (_lib.WgradProblem * n)()
        keep = []
        self.flops = 0.0
        for i, (dY, X, dW, db) in enumerate(problems):
            _dev(dY, X, dW, db)
            assert is16(dY.dtype) and X.dtype == dY.dtype and dW.dtype == torch.float32
            assert i == 0 or dY.dtype == problems[0][0].dtype, "one element type per grouped launch"
            assert dY.shape[0] >= tokens_padded and X.shape[0] >= tokens_padded
            q = arr[i]
            q.dY, q.ld_dy = dY.data_ptr(), dY.stride(0)
            q.X, q.ld_x = X.data_ptr(), X.stride(0)
            q.dW, q.ld_dw = dW.data_ptr(), dW.stride(0)
            q.dbias = _p(db)
            q.n_out, q.n_in = dW.shape
            keep.append((dY, X, dW, db))
            self.flops += 2.0 * tokens * dW.shape[0] * dW.shape[1]
        self._arr, self._keep, self._n = arr, keep, n
        self._tokens, self._tpad = tokens, tokens_padded
        self._dt = dt(problems[0][0])
        self._fn = lib().w2v2_wgrad_grouped
        # mirrors the host's choice in csrc/wgrad.hip (for the in-step profile labels): 256x256 tiles when they alone
        # fill >= 80 % of the CUs and cost fewer rounds than 256x128 tiles -- then the phased kernel
        cd = lambda a, b: -(-a // b)
        t4 = sum(cd(dw.shape[0], 256) * cd(dw.shape[1], 256) for _, _, dw, _ in problems)
        t3 = sum(cd(dw.shape[0], 256) * cd(dw.shape[1], 128) for _, _, dw, _ in problems)
        big = max(dw.shape[0] for _, _, dw, _ in problems) > 128
        tiles256 = big and t4 * 10 >= 256 * 8 and cd(t4, 256) * 17 <= cd(t3, 256) * 10
        self.kernel_name = ("wgrad_grouped_phased_kernel" if tiles256 and not os.environ.get("W2V2_NO_WGRAD_PH") else
                            "wgrad_grouped_ring4_kernel" if tiles256 else
                            "wgrad_grouped_ring_kernel" if big else "wgrad_grouped_kernel")

    def __call__(self) -> None:
        if Gemm._log is not None:
            Gemm._log.append({"kind": "wgrad", "problems": self._n, "tokens": self._tokens, "flops": self.flops,
                              "alg_flops": self.flops,
                              "kernel": self.kernel_name})
        rc = self._fn(self._arr, self._n, self._tokens, self._tpad, self._dt, stream())
        if rc:
            _lib.check(rc, "wgrad_grouped")


# ------------------------------------------------------------------------------------------------ conv0
def conv0_workspace(B: int, N: int, C: int, k: int, stride: int, device) -> torch.Tensor:
    """f32 scratch for conv0_groupnorm_gelu: per-chunk partial sums + {mean, rstd}."""
    n = lib().w2v2_conv0_workspace_floats(N, C, k, stride)
    return torch.empty(B * n + B * C * 2, dtype=torch.float32, device=device)


def conv0_groupnorm_gelu(wav: torch.Tensor, w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                         out: torch.Tensor, work: torch.Tensor, k: int, stride: int, eps: float = 1e-5) -> None:
    """HF:302-323 layer 0.  wav [B,N] f32, w [C,1,k] f32, out [B,L,C], work from conv0_workspace()."""
    _dev(wav, w, gamma, beta, out, work)
    B, N = wav.shape
    Cc = w.shape[0]
    mr = work[work.numel() - B * Cc * 2:]
    L = lib()
    # 16-bit activations: conv on the matrix cores (split-bf16, ~2^-16); statistics from the waveform's window moments
    stats = L.w2v2_conv0_stats_mfma if is16(out.dtype) else L.w2v2_conv0_stats
    _lib.check(stats(wav.data_ptr(), w.data_ptr(), work.data_ptr(), mr.data_ptr(), B, N, Cc, k, stride,
                     eps, stream()), "conv0_stats")
    _lib.check(L.w2v2_conv0_apply(wav.data_ptr(), w.data_ptr(), mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                  out.data_ptr(), dt(out), B, N, Cc, k, stride, stream()), "conv0_apply")


def conv0_layernorm_gelu(wav, w, bias, gamma, beta, out, k: int, stride: int, eps: float = 1e-5) -> None:
    """Layer 0 of the feat_extract_norm="layer" family: out [B, L, C] = GELU(LN_C(conv0(wav) + bias)) (HF:275-299)."""
    _dev(wav, w, bias, gamma, beta, out)
    B, N = wav.shape
    _lib.check(lib().w2v2_conv0_layernorm_gelu(wav.data_ptr(), w.data_ptr(), _p(bias), gamma.data_ptr(), beta.data_ptr(),
                                               out.data_ptr(), B, N, out.shape[-1], k, stride, eps, dt(out), stream()),
               "conv0_layernorm_gelu")


def conv0_bwd(wav, w, work, gamma, beta, dz, sums, dw, dgamma, dbeta, k: int, stride: int) -> None:
    """Backward of layer 0 (see w2v2_conv0_bwd); `work` is the forward's conv0 workspace (holds mean/rstd)."""
    _dev(wav, w, work, gamma, beta, dz, sums, dw, dgamma, dbeta)
    B, N = wav.shape
    Cc = w.shape[0]
    mr = work[work.numel() - B * Cc * 2:]
    _lib.check(lib().w2v2_conv0_bwd(wav.data_ptr(), w.data_ptr(), mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    dz.data_ptr(), sums.data_ptr(), dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                    dt(dz), B, N, Cc, k, stride, stream()), "conv0_bwd")


def col2im(col, dx, B: int, Lin: int, Lout: int, Cin: int, k: int, stride: int) -> None:
    _dev(col, dx)
    _lib.check(lib().w2v2_col2im(col.data_ptr(), dx.data_ptr(), B, Lin, Lout, Cin, k, stride, dt(col), stream()),
               "col2im")


def unpack_conv_grad(gp, g) -> None:
    _dev(gp, g)
    co, ci, k = g.shape
    _lib.check(lib().w2v2_unpack_conv_grad(gp.data_ptr(), g.data_ptr(), co, ci, k, stream()), "unpack_conv_grad")


def pack_conv_weight(w: torch.Tensor, out: torch.Tensor) -> None:
    _dev(w, out)
    co, ci, k = w.shape
    _lib.check(lib().w2v2_pack_conv_weight(w.data_ptr(), out.data_ptr(), dt(out), co, ci, k, stream()), "pack_conv")


# ------------------------------------------------------------------------------------------------ norm
def layernorm_fwd(x, r, gamma, beta, y, mean, rstd, eps: float, drop_p: float = 0.0, seed: int = 0) -> None:
    _dev(x, r, gamma, beta, y, mean, rstd)
    H = x.shape[-1]
    M = x.numel() // H
    _lib.check(lib().w2v2_layernorm_fwd(x.data_ptr(), _p(r), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                        mean.data_ptr(), rstd.data_ptr(), M, H, eps, drop_p, seed, dt(x), stream()),
               "layernorm_fwd")


def layernorm_gelu_fwd(x, gamma, beta, y, eps: float) -> None:
    """y = GELU(LN(x)): the convolution layers of the feat_extract_norm="layer" family (HF:275-299)."""
    _dev(x, gamma, beta, y)
    H = x.shape[-1]
    _lib.check(lib().w2v2_layernorm_gelu_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.numel() // H, H,
                                             eps, dt(x), stream()), "layernorm_gelu_fwd")


_LN_WS = {}


def _ln_workspace(H: int, device) -> torch.Tensor:
    key = (H, str(device))
    if key not in _LN_WS:
        _LN_WS[key] = torch.empty(lib().w2v2_layernorm_bwd_workspace_floats(H), dtype=torch.float32, device=device)
    return _LN_WS[key]


def layernorm_bwd(dy, s, mean, rstd, gamma, ds, d_r, dgamma, dbeta, drop_p: float = 0.0, seed: int = 0,
                  defer_to: Optional["LnFoldGroup"] = None) -> None:
    """With ``defer_to`` the column partials stay in a workspace of that group and dgamma/dbeta are folded later by
    ``defer_to.fold()`` (one launch for up to 8 LayerNorms)."""
    _dev(dy, s, mean, rstd, gamma, ds, d_r, dgamma, dbeta)
    H = dy.shape[-1]
    M = dy.numel() // H
    if defer_to is not None and dgamma is not None:
        ws = defer_to.add(dgamma, dbeta, M, H)
        _lib.check(lib().w2v2_layernorm_bwd(dy.data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            gamma.data_ptr(), ds.data_ptr(), _p(d_r), None, None, ws.data_ptr(), M, H,
                                            drop_p, seed, dt(dy), stream()), "layernorm_bwd")
        return
    ws = _ln_workspace(H, dy.device) if dgamma is not None else None
    _lib.check(lib().w2v2_layernorm_bwd(dy.data_ptr(), s.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        gamma.data_ptr(), ds.data_ptr(), _p(d_r), _p(dgamma), _p(dbeta), _p(ws), M, H,
                                        drop_p, seed, dt(dy), stream()), "layernorm_bwd")


class LnFoldGroup:
    """Pending LayerNorm gamma/beta folds of one gradient bucket: each deferred layernorm_bwd gets its own workspace
    slot (allocated once), fold() issues ONE w2v2_layernorm_bwd_fold launch for all of them."""

    def __init__(self, H: int, device, slots: int = 8):
        n = lib().w2v2_layernorm_bwd_workspace_floats(H)
        self._ws = torch.empty(slots, n, dtype=torch.float32, device=device)
        self._arr = (_lib.LnFold * slots)()
        self._n, self._M, self._H, self._keep = 0, 0, H, []

    def add(self, dgamma, dbeta, M: int, H: int) -> torch.Tensor:
        if self._n == len(self._arr):
            self.fold()
        assert H == self._H and (self._n == 0 or M == self._M)
        e = self._arr[self._n]
        e.partial, e.dgamma, e.dbeta = self._ws[self._n].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
        self._keep.append((dgamma, dbeta))
        self._M = M
        self._n += 1
        return self._ws[self._n - 1]

    def fold(self) -> None:
        if self._n:
            _lib.check(lib().w2v2_layernorm_bwd_fold(self._arr, self._n, self._M, self._H, stream()),
                       "layernorm_bwd_fold")
        self._n, self._keep = 0, []


# ------------------------------------------------------------------------------------------------ elementwise
def dropout_(x: torch.Tensor, p: float, seed: int) -> None:
    _dev(x)
    _lib.check(lib().w2v2_dropout(x.data_ptr(), x.data_ptr(), x.numel(), p, seed, dt(x), stream()), "dropout")


def gelu_bwd(dy, pre, dx) -> None:
    _dev(dy, pre, dx)
    _lib.check(lib().w2v2_gelu_bwd(dy.data_ptr(), pre.data_ptr(), dx.data_ptr(), dy.numel(), dt(dy), stream()),
               "gelu_bwd")


def gelu_bwd_colsum(dy, pre, dx, colsum_out, M: int, N: int) -> None:
    """dx = dy * gelu'(pre) over [M, N] and colsum_out[n] += sum_m dx[m, n] in one pass."""
    _dev(dy, pre, dx, colsum_out)
    _lib.check(lib().w2v2_gelu_bwd_colsum(dy.data_ptr(), pre.data_ptr(), dx.data_ptr(), colsum_out.data_ptr(), M, N,
                                          dt(dy), stream()), "gelu_bwd_colsum")


def add(x, a, y) -> None:
    _dev(x, a, y)
    _lib.check(lib().w2v2_add(x.data_ptr(), a.data_ptr(), y.data_ptr(), x.numel(), dt(x), stream()), "add")


def colsum(x: torch.Tensor, out: torch.Tensor, M: int, N: int, ld: Optional[int] = None) -> None:
    """out[n] += sum_m x[m][n] (f32 atomics)."""
    _dev(x, out)
    _lib.check(lib().w2v2_colsum(x.data_ptr(), ld if ld is not None else N, out.data_ptr(), M, N, dt(x), stream()),
               "colsum")


def zero_ranges(base: torch.Tensor, table: torch.Tensor, blocks_per_range: int = 8) -> None:
    """base[off : off + n] = 0 for every (off, n) row of the int64 device table (f32 arena)."""
    _dev(base, table)
    assert base.dtype == torch.float32 and table.dtype == torch.int64
    _lib.check(lib().w2v2_zero_ranges(base.data_ptr(), table.data_ptr(), table.shape[0], blocks_per_range, stream()),
               "zero_ranges")


def mean(x: torch.Tensor, out: torch.Tensor) -> None:
    _dev(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.float32
    _lib.check(lib().w2v2_mean(x.data_ptr(), out.data_ptr(), x.numel(), stream()), "mean")


def cast(x: torch.Tensor, y: torch.Tensor) -> None:
    _dev(x, y)
    _lib.check(lib().w2v2_cast(x.data_ptr(), y.data_ptr(), x.numel(), dt(y), stream()), "cast")


def transpose_many(src, dst, table, n: int, blocks_per_matrix: int = 64) -> None:
    _dev(src, dst, table)
    _lib.check(lib().w2v2_transpose_many(src.data_ptr(), dst.data_ptr(), table.data_ptr(), n, blocks_per_matrix,
                                         dt(src), stream()), "transpose_many")


def mask_fill(h, mask_u8, embed) -> None:
    _dev(h, mask_u8, embed)
    H = h.shape[-1]
    _lib.check(lib().w2v2_mask_fill(h.data_ptr(), mask_u8.data_ptr(), embed.data_ptr(), h.numel() // H, H, dt(h),
                                    stream()), "mask_fill")


def mask_feature(h, mask_u8, B: int, T: int) -> None:
    """h [B*T, H] (or [B, T, H]): channels c with mask_u8[b][c] != 0 are zeroed for every frame of utterance b."""
    _dev(h, mask_u8)
    H = h.shape[-1]
    assert mask_u8.dtype == torch.uint8 and mask_u8.numel() == B * H
    _lib.check(lib().w2v2_mask_feature(h.data_ptr(), mask_u8.data_ptr(), B, T, H, dt(h), stream()), "mask_feature")


def mask_fill_bwd(dh, mask_u8, d_embed) -> None:
    _dev(dh, mask_u8, d_embed)
    H = dh.shape[-1]
    _lib.check(lib().w2v2_mask_fill_bwd(dh.data_ptr(), mask_u8.data_ptr(), d_embed.data_ptr(), dh.numel() // H, H,
                                        dt(dh), stream()), "mask_fill_bwd")


def prepend_token(x, y, c: float) -> None:
    _dev(x, y)
    B, T, H = x.shape
    _lib.check(lib().w2v2_prepend_token(x.data_ptr(), y.data_ptr(), c, B, T, H, dt(x), stream()), "prepend_token")


# ------------------------------------------------------------------------------------------------ pos-conv
def posconv_wgrad(dY, xg, dwf, B: int, T: int, H: int, G: int, K: int) -> None:
    """dwf[g][(tap, ci)][co] = sum_{b,t} xg[b,g,t+tap,ci] * dY[b,t,g*Cg+co]  (bf16 operands, f32 result, overwritten)."""
    _dev(dY, xg, dwf)
    assert is16(dY.dtype) and xg.dtype == dY.dtype and dwf.dtype == torch.float32
    _lib.check(lib().w2v2_posconv_wgrad(dY.data_ptr(), xg.data_ptr(), dwf.data_ptr(), B, T, H, G, K, dt(dY), stream()),
               "posconv_wgrad")


def posconv_direct(xg, w, out, aux, bias, B: int, T: int, G: int, Cg: int, K: int, ldc: int, mode: int) -> None:
    """Direct grouped pos-conv / its data gradient (include/w2v2_hip.h w2v2_posconv_direct): mode 0 = GELU(conv + bias)
    (+ pre-activation into aux), mode 1 = conv + aux."""
    _dev(xg, w, out, aux, bias)
    assert is16(xg.dtype) and w.dtype == xg.dtype and out.dtype == xg.dtype
    _lib.check(lib().w2v2_posconv_direct(xg.data_ptr(), w.data_ptr(), out.data_ptr(), _p(aux), _p(bias), B, T, G, Cg, K,
                                         ldc, mode, dt(xg), stream()), "posconv_direct")
    if Gemm._log is not None:
        Gemm._log.append({"kind": "posconv", "flops": 2.0 * B * T * G * Cg * K * Cg, "alg_flops": 2.0 * B * T * G * Cg * K * Cg,
                          "kernel": "posconv_direct_kernel", "M": B * T, "N": Cg, "K": K * Cg, "mode": mode})


def posconv_regroup(x, xg, B: int, T: int, H: int, G: int, K: int, pad_left: int) -> None:
    _dev(x, xg)
    _lib.check(lib().w2v2_posconv_regroup(x.data_ptr(), xg.data_ptr(), B, T, H, G, K, pad_left, dt(x), stream()),
               "posconv_regroup")


def weightnorm_scratch(H: int, G: int, K: int, device) -> torch.Tensor:
    """f32 scratch of weightnorm_pack (``sumsq``) / weightnorm_bwd (``dot``): per-tap result + per-block partials."""
    return torch.empty(int(lib().w2v2_weightnorm_scratch_floats(H, G, K)), dtype=torch.float32, device=device)


def weightnorm_pack(g, v, sumsq, wf, wb, H: int, G: int, K: int) -> None:
    _dev(g, v, sumsq, wf, wb)
    _lib.check(lib().w2v2_weightnorm_pack(g.data_ptr(), v.data_ptr(), sumsq.data_ptr(), wf.data_ptr(),
                                          wb.data_ptr(), H, G, K, dt(wf), stream()), "weightnorm_pack")


def weightnorm_bwd(g, v, sumsq, dwf, dot, dg, dv, H: int, G: int, K: int) -> None:
    _dev(g, v, sumsq, dwf, dot, dg, dv)
    _lib.check(lib().w2v2_weightnorm_bwd(g.data_ptr(), v.data_ptr(), sumsq.data_ptr(), dwf.data_ptr(),
                                         dot.data_ptr(), dg.data_ptr(), dv.data_ptr(), H, G, K, stream()),
               "weightnorm_bwd")


# ------------------------------------------------------------------------------------------------ attention
def softmax_fwd(s, p, p_drop, rows: int, T: int, ld: int, drop_p: float, seed: int) -> None:
    _dev(s, p, p_drop)
    _lib.check(lib().w2v2_softmax_fwd(s.data_ptr(), p.data_ptr(), _p(p_drop), rows, T, ld, drop_p, seed, dt(p),
                                      stream()), "softmax_fwd")


def softmax_bwd(dp_drop, p, ds, rows: int, T: int, ld: int, drop_p: float, seed: int) -> None:
    _dev(dp_drop, p, ds)
    _lib.check(lib().w2v2_softmax_bwd(dp_drop.data_ptr(), p.data_ptr(), ds.data_ptr(), rows, T, ld, drop_p, seed,
                                      dt(p), stream()), "softmax_bwd")


def attention_fwd(qkv, ctx, lse, B: int, T: int, heads: int, d: int, scale: float, drop_p: float, seed: int) -> None:
    _dev(qkv, ctx, lse)
    _lib.check(lib().w2v2_attention_fwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, T, heads, d, scale,
                                        drop_p, seed, dt(qkv), stream()), "attention_fwd")


def attention_bwd(qkv, ctx, dctx, lse, dqkv, delta, B: int, T: int, heads: int, d: int, scale: float,
                  drop_p: float, seed: int) -> None:
    _dev(qkv, ctx, dctx, lse, dqkv, delta)
    _lib.check(lib().w2v2_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(),
                                        dqkv.data_ptr(), delta.data_ptr(), B, T, heads, d, scale, drop_p, seed,
                                        dt(qkv), stream()), "attention_bwd")


# ------------------------------------------------------------------------------------------------ pooling
def pool_fwd(x, out, mode: int) -> None:
    _dev(x, out)
    B, T, H = x.shape
    _lib.check(lib().w2v2_pool_fwd(x.data_ptr(), out.data_ptr(), B, T, H, mode, dt(x), stream()), "pool_fwd")


def pool_bwd(x, out, dout, dx, mode: int) -> None:
    _dev(x, out, dout, dx)
    B, T, H = x.shape
    _lib.check(lib().w2v2_pool_bwd(x.data_ptr(), out.data_ptr(), dout.data_ptr(), dx.data_ptr(), B, T, H, mode,
                                   dt(x), stream()), "pool_bwd")


# ------------------------------------------------------------------------------------------------ heads
def row_invnorm(x, inv, rows: int, cols: int, ld: Optional[int] = None) -> None:
    _dev(x, inv)
    _lib.check(lib().w2v2_row_invnorm(x.data_ptr(), ld if ld is not None else cols, inv.data_ptr(), rows, cols,
                                      dt(x), stream()), "row_invnorm")


def aam_softmax_fwd_bwd(cos, label, softmax, loss_rows, dcos_w, dcos_x, inv_x, inv_w, rowdot, colprod, B: int,
                        Cn: int, ldc: int, margin: float, scale: float, loss_scale=None, correct_rows=None,
                        easy_margin: bool = False) -> None:
    """loss_scale: device tensor whose first element multiplies the loss gradient (fp16 loss scaling), or None.
    correct_rows [B] f32 (optional): 1 where the arg-max prediction equals the label (train_acc).
    colprod [B, Cn] f32 (optional): g * cos per element; colsum() over its rows gives the column dots."""
    _dev(cos, label, softmax, loss_rows, dcos_w, dcos_x, inv_x, inv_w, rowdot, colprod, loss_scale, correct_rows)
    dty = dt(dcos_w) if dcos_w is not None else F32
    _lib.check(lib().w2v2_aam_softmax_fwd_bwd(cos.data_ptr(), label.data_ptr(), softmax.data_ptr(),
                                              loss_rows.data_ptr(), _p(dcos_w), _p(dcos_x), _p(inv_x), _p(inv_w),
                                              _p(rowdot), _p(colprod), B, Cn, ldc, margin, scale, _p(loss_scale),
                                              _p(correct_rows), int(easy_margin), dty, stream()),
               "aam_softmax")


def normalize_bwd(g, x, inv, dot, dx, rows: int, cols: int, ldx: Optional[int] = None, add_to: bool = False) -> None:
    _dev(g, x, inv, dot, dx)
    _lib.check(lib().w2v2_normalize_bwd(g.data_ptr(), x.data_ptr(), ldx if ldx is not None else cols, inv.data_ptr(),
                                        dot.data_ptr(), dx.data_ptr(), rows, cols, dt(x), int(add_to), stream()),
               "normalize_bwd")


def aam_dw(dcos_x, emb, colprod, w_master, inv_w, dw, B: int, Cn: int, E: int, ldc: int) -> None:
    """Class-weight gradient of the AAM head (product + column dots + F.normalize backward) in one launch."""
    _dev(dcos_x, emb, colprod, w_master, inv_w, dw)
    assert emb.dtype == dcos_x.dtype and dw.dtype == torch.float32 and w_master.dtype == torch.float32
    _lib.check(lib().w2v2_aam_dw(dcos_x.data_ptr(), ldc, emb.data_ptr(), colprod.data_ptr(), w_master.data_ptr(),
                                 inv_w.data_ptr(), dw.data_ptr(), B, Cn, E, dt(emb), stream()), "aam_dw")


# ------------------------------------------------------------------------------------------------ optimiser
def adam_step(p, g, m, v, pb, n: int, lr: float, beta1: float, beta2: float, eps: float, step: int,
              grad_scale: float = 1.0, scaler=None, skip_slot: int = 0) -> None:
    """scaler: the device record of the dynamic loss scale (see grad_scaler_*), or None.  skip_slot (4 = head range,
    5 = body range of an 8-float record): the kernel takes the bias corrections at step - scaler[skip_slot], i.e. the
    number of optimiser steps that were NOT skipped for an overflow (torch GradScaler semantics)."""
    _dev(p, g, m, v, pb, scaler)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _lib.check(lib().w2v2_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(pb),
                                    dt(pb) if pb is not None else BF16, n, lr, beta1, beta2, eps, bc1, bc2, grad_scale,
                                    _p(scaler), step, skip_slot, stream()), "adam_step")


def weight_residual(p, lo, table) -> None:
    """lo[off:off+n] = T(p[off:off+n] - T(p[off:off+n])) for every (off, n) row of the int64 device table."""
    _dev(p, lo, table)
    _lib.check(lib().w2v2_weight_residual(p.data_ptr(), lo.data_ptr(), table.data_ptr(), table.shape[0], dt(lo), stream()),
               "weight_residual")


def grad_scaler_check(g, n: int, state) -> None:
    """state[1] = 1 if any of g[:n] is non-finite (torch GradScaler's unscale_ / found_inf)."""
    _dev(g, state)
    _lib.check(lib().w2v2_grad_scaler_check(g.data_ptr(), n, state.data_ptr(), stream()), "grad_scaler_check")


def grad_scaler_update(state, growth: float = 2.0, backoff: float = 0.5, growth_interval: int = 2000,
                       skipped_ranges: int = 0) -> None:
    """torch GradScaler.update() on the device record {scale, found_inf, growth_tracker, skipped_steps [, skipped_head,
    skipped_body, -, -]}; skipped_ranges bit 0 / 1: this step covered the head / body range (8-float records)."""
    _dev(state)
    assert skipped_ranges == 0 or state.numel() >= 8
    _lib.check(lib().w2v2_grad_scaler_update(state.data_ptr(), growth, backoff, growth_interval, skipped_ranges, stream()),
               "grad_scaler_update")


# ------------------------------------------------------------------------------------------------ attentive pooling
def asp_context(x, ctx, B: int, T: int, C: int) -> None:
    _dev(x, ctx)
    _lib.check(lib().w2v2_asp_context(x.data_ptr(), ctx.data_ptr(), B, T, C, dt(x), stream()), "asp_context")


def asp_context_bias(ctx, w1, b1, cb, B: int, A: int, C: int) -> None:
    _dev(ctx, w1, b1, cb)
    _lib.check(lib().w2v2_asp_context_bias(ctx.data_ptr(), w1.data_ptr(), b1.data_ptr(), cb.data_ptr(), B, A, C,
                                           stream()), "asp_context_bias")


def asp_bn_workspace(M: int, A: int, device) -> torch.Tensor:
    return torch.empty(lib().w2v2_asp_bn_workspace_floats(M, A), dtype=torch.float32, device=device)


def asp_bn_stats(a_pre, work, mean_rstd, running, M: int, A: int, eps: float, momentum: float) -> None:
    _dev(a_pre, work, mean_rstd, running)
    _lib.check(lib().w2v2_asp_bn_stats(a_pre.data_ptr(), work.data_ptr(), mean_rstd.data_ptr(), _p(running), M, A, eps,
                                       momentum, dt(a_pre), stream()), "asp_bn_stats")


def asp_bn_eval_stats(running, mean_rstd, A: int, eps: float) -> None:
    _dev(running, mean_rstd)
    _lib.check(lib().w2v2_asp_bn_eval_stats(running.data_ptr(), mean_rstd.data_ptr(), A, eps, stream()),
               "asp_bn_eval_stats")


def asp_bn_tanh(a_pre, mean_rstd, gamma, beta, h, M: int, A: int) -> None:
    _dev(a_pre, mean_rstd, gamma, beta, h)
    _lib.check(lib().w2v2_asp_bn_tanh(a_pre.data_ptr(), mean_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      h.data_ptr(), M, A, dt(a_pre), stream()), "asp_bn_tanh")


def asp_bn_bwd(dh, a_pre, mean_rstd, gamma, beta, work, dgamma, dbeta, da, M: int, A: int) -> None:
    _dev(dh, a_pre, mean_rstd, gamma, beta, work, dgamma, dbeta, da)
    _lib.check(lib().w2v2_asp_bn_bwd(dh.data_ptr(), a_pre.data_ptr(), mean_rstd.data_ptr(),
                                     gamma.data_ptr(), beta.data_ptr(), work.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                     da.data_ptr(), M, A, dt(dh), stream()), "asp_bn_bwd")


def asp_pool_fwd(x, s, out, stats, B: int, T: int, C: int) -> None:
    _dev(x, s, out, stats)
    _lib.check(lib().w2v2_asp_pool_fwd(x.data_ptr(), s.data_ptr(), out.data_ptr(), stats.data_ptr(), B, T, C, dt(x),
                                       stream()), "asp_pool_fwd")


def asp_pool_bwd(x, s, out, stats, dout, ds, dx, B: int, T: int, C: int) -> None:
    _dev(x, s, out, stats, dout, ds, dx)
    _lib.check(lib().w2v2_asp_pool_bwd(x.data_ptr(), s.data_ptr(), out.data_ptr(), stats.data_ptr(), dout.data_ptr(),
                                       ds.data_ptr(), dx.data_ptr(), B, T, C, dt(x), stream()), "asp_pool_bwd")


def asp_context_bwd(x, ctx, da, w1, dw1, dx, scratch, B: int, T: int, C: int, A: int) -> None:
    _dev(x, ctx, da, w1, dw1, dx, scratch)
    _lib.check(lib().w2v2_asp_context_bwd(x.data_ptr(), ctx.data_ptr(), da.data_ptr(), w1.data_ptr(), dw1.data_ptr(),
                                          dx.data_ptr(), scratch.data_ptr(), B, T, C, A, dt(x), stream()),
               "asp_context_bwd")


# ------------------------------------------------------------------------------------------------ ECAPA-TDNN pieces
def bn_workspace(M: int, C: int, device) -> torch.Tensor:
    return torch.empty(lib().w2v2_bn_workspace_floats(M, C), dtype=torch.float32, device=device)


def bn_fwd(a, lda: int, work, mean_rstd, running, gamma, beta, y, ldy: int, M: int, C: int, eps: float,
           momentum: float, relu: bool, train: bool) -> None:
    """BatchNorm1d (optionally on relu(a)): batch statistics + running update (train) or the running statistics."""
    _dev(a, work, mean_rstd, running, gamma, beta, y)
    _lib.check(lib().w2v2_bn_fwd(a.data_ptr(), lda, _p(work), mean_rstd.data_ptr(), _p(running), gamma.data_ptr(),
                                 beta.data_ptr(), y.data_ptr(), ldy, M, C, eps, momentum, int(relu), int(train), dt(a),
                                 stream()), "bn_fwd")


def bn_colsum_rows(M: int, C: int) -> int:
    return int(lib().w2v2_bn_colsum_rows(M, C))


def bn_bwd(dy, lddy: int, a, lda: int, mean_rstd, gamma, work, dgamma, dbeta, da, ldda: int, M: int, C: int,
           relu: bool, colsum_partial=None, dy2=None, lddy2: int = 0) -> None:
    """colsum_partial [bn_colsum_rows(M, C), C] f32 (optional): written with the per-row-block column sums of da.
    dy2 (optional, same shape as dy, own row stride): the output gradient is dy + dy2."""
    _dev(dy, a, mean_rstd, gamma, work, dgamma, dbeta, da, colsum_partial, dy2)
    if dy2 is None:
        _lib.check(lib().w2v2_bn_bwd(dy.data_ptr(), lddy, a.data_ptr(), lda, mean_rstd.data_ptr(), gamma.data_ptr(),
                                     work.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), da.data_ptr(), ldda, M, C,
                                     int(relu), _p(colsum_partial), dt(a), stream()), "bn_bwd")
    else:
        _lib.check(lib().w2v2_bn_bwd_sum(dy.data_ptr(), lddy, dy2.data_ptr(), lddy2, a.data_ptr(), lda,
                                         mean_rstd.data_ptr(), gamma.data_ptr(), work.data_ptr(), dgamma.data_ptr(),
                                         dbeta.data_ptr(), da.data_ptr(), ldda, M, C, int(relu), _p(colsum_partial),
                                         dt(a), stream()), "bn_bwd_sum")


def im2col_reflect(x, ldx: int, col, B: int, T: int, Cin: int, k: int, dilation: int, x2=None, ldx2: int = 0) -> None:
    """x2 (optional, same shape, own row stride): the taps of x + x2."""
    _dev(x, col, x2)
    if x2 is None:
        _lib.check(lib().w2v2_im2col_reflect(x.data_ptr(), ldx, col.data_ptr(), B, T, Cin, k, dilation, dt(x), stream()),
                   "im2col_reflect")
    else:
        _lib.check(lib().w2v2_im2col_reflect_sum(x.data_ptr(), ldx, x2.data_ptr(), ldx2, col.data_ptr(), B, T, Cin, k,
                                                 dilation, dt(x), stream()), "im2col_reflect_sum")


def col2im_reflect(dcol, dx, lddx: int, B: int, T: int, Cin: int, k: int, dilation: int, accumulate: bool) -> None:
    _dev(dcol, dx)
    _lib.check(lib().w2v2_col2im_reflect(dcol.data_ptr(), dx.data_ptr(), lddx, B, T, Cin, k, dilation,
                                         int(accumulate), dt(dcol), stream()), "col2im_reflect")


def add_strided(a, lda: int, b, ldb: int, y, ldy: int, M: int, C: int) -> None:
    """y = a + b over row-strided [M, C] views; b None: y = a (strided copy)."""
    _dev(a, b, y)
    _lib.check(lib().w2v2_add_strided(a.data_ptr(), lda, _p(b), ldb, y.data_ptr(), ldy, M, C, dt(a), stream()),
               "add_strided")


def copy_strided(a, lda: int, y, ldy: int, M: int, C: int) -> None:
    add_strided(a, lda, None, 0, y, ldy, M, C)


def se_scale(x, g, y, B: int, T: int, C: int) -> None:
    _dev(x, g, y)
    _lib.check(lib().w2v2_se_scale(x.data_ptr(), g.data_ptr(), y.data_ptr(), B, T, C, dt(x), stream()), "se_scale")


def se_bwd_gate(dout, x, dg, B: int, T: int, C: int) -> None:
    _dev(dout, x, dg)
    _lib.check(lib().w2v2_se_bwd_gate(dout.data_ptr(), x.data_ptr(), dg.data_ptr(), B, T, C, dt(x), stream()),
               "se_bwd_gate")


def se_bwd_x(dout, g, ds, dx, B: int, T: int, C: int) -> None:
    _dev(dout, g, ds, dx)
    _lib.check(lib().w2v2_se_bwd_x(dout.data_ptr(), g.data_ptr(), ds.data_ptr(), dx.data_ptr(), B, T, C, dt(dout),
                                   stream()), "se_bwd_x")


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def skinny_linear_fwd(x, W, bias, y, act: int) -> None:
    """y [B, N] = act(x [B, K] @ W [N, K]^T + bias) in exact f32 (csrc/skinny.hip)."""
    _dev(x, W, bias, y)
    assert x.dtype == W.dtype == y.dtype == torch.float32 and x.is_contiguous() and W.is_contiguous() and y.is_contiguous()
    B, K = x.shape
    N = W.shape[0]
    assert W.numel() == N * K and y.shape == (B, N)
    _lib.check(lib().w2v2_skinny_linear_fwd(x.data_ptr(), W.data_ptr(), _p(bias), y.data_ptr(), B, N, K, act, stream()),
               "skinny_linear_fwd")


def skinny_linear_bwd_x(dy, y, W, dx, act: int) -> None:
    """dx [B, K] = (dy * act'(y)) [B, N] @ W [N, K]."""
    _dev(dy, y, W, dx)
    B, N = dy.shape
    K = dx.shape[1]
    assert W.numel() == N * K and dx.shape == (B, K) and dx.is_contiguous() and dy.is_contiguous()
    _lib.check(lib().w2v2_skinny_linear_bwd_x(dy.data_ptr(), _p(y), W.data_ptr(), dx.data_ptr(), B, N, K, act, stream()),
               "skinny_linear_bwd_x")


def skinny_linear_bwd_w(dy, y, x, dW, dbias, act: int, accumulate: bool) -> None:
    """dW [N, K] (+)= (dy * act'(y))^T @ x [B, K]; dbias [N] (+)= its column sums."""
    _dev(dy, y, x, dW, dbias)
    B, N = dy.shape
    K = x.shape[1]
    assert dW.numel() == N * K and x.shape == (B, K) and x.is_contiguous() and dy.is_contiguous()
    _lib.check(lib().w2v2_skinny_linear_bwd_w(dy.data_ptr(), _p(y), x.data_ptr(), dW.data_ptr(), _p(dbias), B, N, K, act,
                                              int(accumulate), stream()), "skinny_linear_bwd_w")


def act_fwd(x, y, mode: int) -> None:
    _dev(x, y)
    _lib.check(lib().w2v2_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), mode, stream()), "act_fwd")


def act_bwd(dy, y, dx, mode: int) -> None:
    _dev(dy, y, dx)
    _lib.check(lib().w2v2_act_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), mode, stream()), "act_bwd")


def bce_head_fwd_bwd(emb, w, b, label, prob, loss_rows, dlogit, demb, dw, db, B: int, H: int, loss_scale=None) -> None:
    _dev(emb, w, b, label, prob, loss_rows, dlogit, demb, dw, db, loss_scale)
    _lib.check(lib().w2v2_bce_head_fwd_bwd(emb.data_ptr(), w.data_ptr(), b.data_ptr(), label.data_ptr(), prob.data_ptr(),
                                           loss_rows.data_ptr(), _p(dlogit), _p(demb), _p(dw), _p(db), B, H,
                                           _p(loss_scale), stream()), "bce_head")

