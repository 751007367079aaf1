"""Attentive statistics pooling on the HIP path (BASELINE configs[2]).

ref call site: src/layers/pooling.py:87-106 (``AttentiveStatPool1D`` -> speechbrain 0.5.x
``AttentiveStatisticsPooling(channels, attention_channels=128, global_context=True)``).  speechbrain is not part
of the reference tree and is not installed: the arithmetic follows its published definition, restated in
``oracle.attentive_stat_pool`` -- parity for this row is against that restatement only ("parity unpinned").

Parameters live in the ParamStore arena right after the classifier (same gradient bucket / Adam slice), under the
speechbrain state-dict names ``stat_pooling.pooling_layer.{tdnn.conv.conv, tdnn.norm.norm, conv.conv}.*``.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import EPI_ADD, EPI_BIAS, Gemm, WgradGroup

ASP_PREFIX = "stat_pooling.pooling_layer."
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def asp_param_shapes(channels: int, attention_channels: int = 128):
    C, A = channels, attention_channels
    return {ASP_PREFIX + "tdnn.conv.conv.weight": (A, 3 * C, 1), ASP_PREFIX + "tdnn.conv.conv.bias": (A,),
            ASP_PREFIX + "tdnn.norm.norm.weight": (A,), ASP_PREFIX + "tdnn.norm.norm.bias": (A,),
            ASP_PREFIX + "conv.conv.weight": (C, A, 1), ASP_PREFIX + "conv.conv.bias": (C,)}


class AttentivePool:
    """x [B*T, C] (the last hidden state, padded storage) -> emb [B, 2C] = [mean_w, std_w]; backward adds the
    gradient of x into ``dx`` and writes the six parameter gradients."""

    def __init__(self, store, x: torch.Tensor, emb: torch.Tensor, dx: torch.Tensor, B: int, T: int, train: bool,
                 prefix: str = ASP_PREFIX):
        self.store, self.B, self.T, self.train = store, B, T, train
        ASP_PREFIX = prefix          # parameter names of this instance (the ECAPA model keeps them under ``asp.``)
        self.prefix = prefix
        C = x.shape[1]
        A = store.shapes[ASP_PREFIX + "tdnn.conv.conv.bias"][0]
        self.C, self.A = C, A
        M = B * T
        dev, adt, f32 = store.device, store.act_dtype, torch.float32
        self.x, self.emb, self.dx = x, emb, dx

        def ep(rows, cols):     # zero-padded to a multiple of 64 rows: legal K-major operand of the grouped wgrad
            full = torch.zeros((rows + 63) // 64 * 64, cols, dtype=adt, device=dev)
            v = full[:rows]
            v._w2v2_padded = full
            return v
        self.ctx = torch.empty(B, 2 * C, dtype=f32, device=dev)
        self.cb = torch.empty(B, A, dtype=f32, device=dev)
        self.a_pre = torch.empty(M, A, dtype=adt, device=dev)
        self.h = ep(M, A)
        self.s = torch.empty(M, C, dtype=adt, device=dev)
        self.mean_rstd = torch.empty(A, 2, dtype=f32, device=dev)
        self.running = store.asp_running                                             # BatchNorm1d buffers {mean, var}
        self.work = ops.asp_bn_workspace(M, A, dev)
        self.stats = torch.empty(B, C, 2, dtype=f32, device=dev)
        p, w = store.p, store.w
        W1 = w(ASP_PREFIX + "tdnn.conv.conv.weight").view(A, 3 * C)
        W2 = w(ASP_PREFIX + "conv.conv.weight").view(C, A)
        # a_pre[b] = x[b] Wx^T + cb[b]: one GEMM batch per utterance so the context term is a per-batch bias
        self.g_a = Gemm(T, A, C, x, W1, self.a_pre, lda=C, ldb=3 * C, ldc=A, batch=B, batch_inner=B,
                        a_strides=(0, T * C), c_strides=(0, T * A), epilogue=EPI_BIAS, bias=self.cb, bias_stride1=A)
        self.g_s = Gemm(M, C, A, self.h, W2, self.s, lda=A, ldb=A, ldc=C, epilogue=EPI_BIAS,
                        bias=p(ASP_PREFIX + "conv.conv.bias"))
        if not train:
            return
        g = store.g
        self.ds, self.da = ep(M, C), ep(M, A)
        self.dh = torch.empty(M, A, dtype=adt, device=dev)
        self.scratch = torch.empty(B * A + B * 2 * C, dtype=f32, device=dev)
        self.g_dh = Gemm(M, A, C, self.ds, W2, self.dh, lda=C, ldb=A, ldc=A, transB=True)
        self.g_dx = Gemm(M, C, A, self.da, W1, dx, lda=A, ldb=3 * C, ldc=C, transB=True, epilogue=EPI_ADD, aux=dx,
                         ldaux=C)
        dW1 = g(ASP_PREFIX + "tdnn.conv.conv.weight").view(A, 3 * C)
        dW2 = g(ASP_PREFIX + "conv.conv.weight").view(C, A)
        self.grouped = ops.is16(adt) and hasattr(x, "_w2v2_padded")
        if self.grouped:
            pad = lambda t: t._w2v2_padded
            self.g_w = WgradGroup([(pad(self.ds), pad(self.h), dW2, g(ASP_PREFIX + "conv.conv.bias")),
                                   (pad(self.da), pad(x), dW1[:, :C], g(ASP_PREFIX + "tdnn.conv.conv.bias"))], M,
                                  pad(self.ds).shape[0])
        else:
            # exact-f32 mode: the token dimension is split over ~2 workgroups per CU (f32 atomics, zeroed target)
            from .ecapa import f32_dw_split
            sk = lambda m, n: f32_dw_split(m, n, M)
            self.g_w2 = Gemm(C, A, M, self.ds, self.h, dW2, lda=C, ldb=A, ldc=A, transA=True, transB=True,
                             split_k=sk(C, A), accumulate=True)
            self.g_w1 = Gemm(A, C, M, self.da, x, dW1, lda=A, ldb=C, ldc=3 * C, transA=True, transB=True,
                             split_k=sk(A, C), accumulate=True)

    def forward(self) -> torch.Tensor:
        st, B, T, C, A = self.store, self.B, self.T, self.C, self.A
        p = st.p
        ops.asp_context(self.x, self.ctx, B, T, C)
        ops.asp_context_bias(self.ctx, p(self.prefix + "tdnn.conv.conv.weight"), p(self.prefix + "tdnn.conv.conv.bias"),
                             self.cb, B, A, C)
        self.g_a()
        if self.train:      # batch statistics (and running-stat update) like BatchNorm1d.train()
            ops.asp_bn_stats(self.a_pre, self.work, self.mean_rstd, self.running, B * T, A, BN_EPS, BN_MOMENTUM)
            if hasattr(st, "asp_batches_tracked"):
                st.asp_batches_tracked += 1
        else:
            ops.asp_bn_eval_stats(self.running, self.mean_rstd, A, BN_EPS)
        ops.asp_bn_tanh(self.a_pre, self.mean_rstd, p(self.prefix + "tdnn.norm.norm.weight"),
                        p(self.prefix + "tdnn.norm.norm.bias"), self.h, B * T, A)
        self.g_s()
        ops.asp_pool_fwd(self.x, self.s, self.emb, self.stats, B, T, C)
        return self.emb

    def backward(self, demb: torch.Tensor) -> None:
        """demb [B, 2C] f32 -> dx (written) + parameter gradients."""
        st, B, T, C, A = self.store, self.B, self.T, self.C, self.A
        p, g = st.p, st.g
        M = B * T
        ops.asp_pool_bwd(self.x, self.s, self.emb, self.stats, demb, self.ds, self.dx, B, T, C)
        self.g_dh()
        ops.asp_bn_bwd(self.dh, self.a_pre, self.mean_rstd, p(self.prefix + "tdnn.norm.norm.weight"),
                       p(self.prefix + "tdnn.norm.norm.bias"), self.work,
                       g(self.prefix + "tdnn.norm.norm.weight"), g(self.prefix + "tdnn.norm.norm.bias"), self.da, M, A)
        if self.grouped:
            self.g_w()
        else:
            self.g_w2()
            ops.colsum(self.ds, g(self.prefix + "conv.conv.bias"), M, C)
            self.g_w1()
            ops.colsum(self.da, g(self.prefix + "tdnn.conv.conv.bias"), M, A)
        self.g_dx()                                                  # dx += da Wx
        ops.asp_context_bwd(self.x, self.ctx, self.da, p(self.prefix + "tdnn.conv.conv.weight"),
                            g(self.prefix + "tdnn.conv.conv.weight"), self.dx, self.scratch, B, T, C, A)
