"""ctypes binding of libw2v2hip.so -- the C ABI declared in include/w2v2_hip.h.

There is NO CPU fallback: if the shared library is missing and cannot be built, importing the ops
raises.  Every wrapper raises ``RuntimeError`` with ``w2v2_last_error()`` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_GELU_BWD, EPI_ADD, EPI_SCALE_RC, EPI_BIAS_GELU_GRAD, EPI_MUL = range(8)

c_i32, c_i64, c_u64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p


class Operand(C.Structure):
    _fields_ = [("ptr", c_vp), ("ld", c_i64), ("seg_len", c_i64), ("seg_stride", c_i64),
                ("stride0", c_i64), ("stride1", c_i64), ("trans", c_i32), ("_pad", c_i32)]


class GemmDesc(C.Structure):
    _fields_ = [("M", c_i32), ("N", c_i32), ("K", c_i32), ("batch", c_i32), ("batch_inner", c_i32),
                ("dtype_ab", c_i32), ("dtype_c", c_i32), ("epilogue", c_i32),
                ("A", Operand), ("B", Operand),
                ("C", c_vp), ("ldc", c_i64), ("c_stride0", c_i64), ("c_stride1", c_i64),
                ("aux", c_vp), ("ldaux", c_i64), ("aux_stride0", c_i64), ("aux_stride1", c_i64),
                ("bias", c_vp), ("bias_stride1", c_i64), ("row_scale", c_vp), ("col_scale", c_vp),
                ("alpha", c_f32), ("split_k", c_i32), ("accumulate", c_i32), ("_pad", c_i32),
                ("k_ext", c_i32), ("n_ext_from", c_i32), ("b_lo_offset", c_i64)]


class LnFold(C.Structure):
    _fields_ = [("partial", c_vp), ("dgamma", c_vp), ("dbeta", c_vp)]


class WgradProblem(C.Structure):
    _fields_ = [("dY", c_vp), ("ld_dy", c_i64), ("X", c_vp), ("ld_x", c_i64), ("dW", c_vp), ("ld_dw", c_i64),
                ("dbias", c_vp), ("n_out", c_i32), ("n_in", c_i32)]


_SIGS = {
    "w2v2_version": (c_i32, []),
    "w2v2_last_error": (C.c_char_p, []),
    "w2v2_gemm": (c_i32, [C.POINTER(GemmDesc), c_vp]),
    "w2v2_gemm_kernel_of": (c_i32, [c_vp]),
    "w2v2_gemm_timed": (c_i32, [C.POINTER(GemmDesc), c_vp, c_i32]),
    "w2v2_timer_read": (c_i32, [c_i32, c_i32, c_vp]),
    "w2v2_tune_gemm_kernel": (c_i32, [c_i32]),
    "w2v2_tune_gemm_debug": (c_i32, [c_i32]),
    "w2v2_tune_gemm_f32_tile": (c_i32, [c_i32]),
    "w2v2_tune_gemm_ring_debug": (c_i32, [c_i32]),
    "w2v2_gemm_f32_last_kernel": (c_i32, []),
    "w2v2_zero_ranges": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "w2v2_mean": (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    "w2v2_wgrad_grouped": (c_i32, [C.POINTER(WgradProblem), c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_tune_wgrad_kernel": (c_i32, [c_i32]),
    "w2v2_conv0_workspace_floats": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "w2v2_conv0_stats": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "w2v2_conv0_stats_mfma": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "w2v2_conv0_apply": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_i32, c_vp]),
    "w2v2_pack_conv_weight": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_conv0_bwd": (c_i32, [c_vp] * 10 + [c_i32] * 6 + [c_vp]),
    "w2v2_col2im": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_unpack_conv_grad": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_conv0_layernorm_gelu": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "w2v2_layernorm_gelu_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "w2v2_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32,
                                   c_u64, c_i32, c_vp]),
    "w2v2_layernorm_bwd_workspace_floats": (c_i32, [c_i32]),
    "w2v2_layernorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                   c_f32, c_u64, c_i32, c_vp]),
    "w2v2_layernorm_bwd_fold": (c_i32, [C.POINTER(LnFold), c_i32, c_i32, c_i32, c_vp]),
    "w2v2_dropout": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_u64, c_i32, c_vp]),
    "w2v2_gelu_bwd_colsum": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_gelu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_add": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_colsum": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_cast": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_transpose_many": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_mask_feature": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_mask_fill": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_mask_fill_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_prepend_token": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_posconv_regroup": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_posconv_direct": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "w2v2_posconv_wgrad": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_weightnorm_scratch_floats": (c_i64, [c_i32, c_i32, c_i32]),
    "w2v2_weightnorm_pack": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_weightnorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_softmax_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_u64, c_i32, c_vp]),
    "w2v2_softmax_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_u64, c_i32, c_vp]),
    "w2v2_attention_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_u64,
                                   c_i32, c_vp]),
    "w2v2_attention_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32,
                                   c_f32, c_u64, c_i32, c_vp]),
    "w2v2_pool_fwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_pool_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_bce_head_fwd_bwd": (c_i32, [c_vp] * 10 + [c_i32, c_i32, c_vp, c_vp]),
    "w2v2_bn_workspace_floats": (c_i32, [c_i32, c_i32]),
    "w2v2_bn_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, c_f32, c_i32,
                            c_i32, c_i32, c_vp]),
    "w2v2_skinny_linear_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_skinny_linear_bwd_x": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_skinny_linear_bwd_w": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_bn_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32,
                            c_vp, c_i32, c_vp]),
    "w2v2_bn_bwd_sum": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32,
                                c_i32, c_i32, c_vp, c_i32, c_vp]),
    "w2v2_bn_colsum_rows": (c_i32, [c_i32, c_i32]),
    "w2v2_im2col_reflect": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_im2col_reflect_sum": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_col2im_reflect": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_add_strided": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_se_scale": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_se_bwd_gate": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_se_bwd_x": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_act_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_act_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_asp_context": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_context_bias": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_bn_workspace_floats": (c_i32, [c_i32, c_i32]),
    "w2v2_asp_bn_stats": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_i32, c_vp]),
    "w2v2_asp_bn_eval_stats": (c_i32, [c_vp, c_vp, c_i32, c_f32, c_vp]),
    "w2v2_asp_bn_tanh": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_bn_bwd": (c_i32, [c_vp] * 9 + [c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_pool_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_pool_bwd": (c_i32, [c_vp] * 7 + [c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_asp_context_bwd": (c_i32, [c_vp] * 7 + [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_row_invnorm": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_aam_softmax_fwd_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                         c_i32, c_i64, c_f32, c_f32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "w2v2_aam_dw": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_normalize_bwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "w2v2_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32,
                               c_f32, c_f32, c_vp, c_i32, c_i32, c_vp]),
    "w2v2_weight_residual": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "w2v2_comm_unique_id": (c_i32, [c_vp]),
    "w2v2_comm_init": (c_i32, [C.POINTER(c_vp), c_vp, c_i32, c_i32, c_i32]),
    "w2v2_comm_init_loopback": (c_i32, [C.POINTER(c_vp), c_i32, c_i32]),
    "w2v2_allreduce_async": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "w2v2_broadcast_async": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "w2v2_comm_destroy": (c_i32, [c_vp]),
    "w2v2_traffic_probe": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_f32, c_vp]),
    "w2v2_grad_scaler_check": (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    "w2v2_grad_scaler_update": (c_i32, [c_vp, c_f32, c_f32, c_i32, c_i32, c_vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def library_path() -> str:
    return _build.LIB


def load():
    """Load (building first if a compiler is present and sources are newer).  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    alt = os.environ.get("W2V2_LIB_AB")        # A/B timing of two builds in one process tree (tools only)
    if alt and os.path.exists(alt):
        path = alt
    elif _build.needs_build():
        # one process per GPU: every rank may get here at once, so the build is serialised on a file lock (the
        # losers find an up-to-date library when they get it).  A failed rebuild is never papered over with a stale
        # library whose ctypes signatures may no longer match: no compiler + sources newer than the .so -> raise.
        try:
            _build.build_locked()
        except Exception as e:
            raise RuntimeError(f"libw2v2hip.so is missing or older than its sources and could not be built: {e}") from e
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        if path == alt and not hasattr(lib, name):
            continue                     # (tools only: an OLDER build under A/B may predate an entry point)
        fn = getattr(lib, name)          # AttributeError if the ABI symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().w2v2_last_error()
        raise RuntimeError(f"libw2v2hip {what} failed: {msg.decode() if msg else rc}")
