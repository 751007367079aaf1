"""Speaker classification heads on the HIP kernels: AAM-softmax (ref: src/optim/loss/aam_softmax.py:21-74)
and Linear + cross-entropy (ref: src/lightning_modules/speaker/wav2vec2_fc.py:199-210,
src/optim/loss/cross_entropy.py:15-33).  One object = fixed buffers + prebuilt GEMM descriptors for a
given (batch, embedding dim, classes); used by engine.Plan and by the nn.Module surface."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .ops import EPI_BIAS, EPI_SCALE_RC, Gemm


class ClassifierHead:
    def __init__(self, kind: str, batch: int, embed_dim: int, classes: int, *, w_master: torch.Tensor,
                 w_operand: torch.Tensor, w_grad: Optional[torch.Tensor], bias: Optional[torch.Tensor] = None,
                 bias_grad: Optional[torch.Tensor] = None, emb: torch.Tensor, act_dtype: torch.dtype,
                 train: bool, margin: float = 0.2, scale: float = 30.0, loss_scale: Optional[torch.Tensor] = None,
                 easy_margin: bool = False):
        assert kind in ("aam", "ce")
        self.loss_scale = loss_scale          # device record of the dynamic loss scale (fp16 activations) or None
        self.kind, self.B, self.E, self.C, self.train = kind, batch, embed_dim, classes, train
        self.margin, self.scale, self.easy_margin = margin, scale, bool(easy_margin)
        self.w_master, self.w_grad, self.bias, self.bias_grad, self.emb = w_master, w_grad, bias, bias_grad, emb
        dev, f32 = emb.device, torch.float32
        B, E, Cn = batch, embed_dim, classes
        self.ldc = (Cn + 7) // 8 * 8
        self.emb_lp = torch.empty(B, E, dtype=act_dtype, device=dev) if act_dtype != f32 else emb
        self.logits = torch.zeros(B, self.ldc, dtype=f32, device=dev)
        self.softmax = torch.zeros(B, self.ldc, dtype=f32, device=dev)
        self.loss_rows = torch.empty(B, dtype=f32, device=dev)
        self.correct = torch.zeros(B, dtype=f32, device=dev)      # 1 where argmax(prediction) == label (train_acc)
        aam = kind == "aam"
        if aam:
            self.inv_x, self.inv_w = torch.empty(B, dtype=f32, device=dev), torch.empty(Cn, dtype=f32, device=dev)
            self.g_fwd = Gemm(B, Cn, E, self.emb_lp, w_operand, self.logits, lda=E, ldb=E, ldc=self.ldc,
                              epilogue=EPI_SCALE_RC, row_scale=self.inv_x, col_scale=self.inv_w)
        else:
            self.g_fwd = Gemm(B, Cn, E, self.emb_lp, w_operand, self.logits, lda=E, ldb=E, ldc=self.ldc,
                              epilogue=EPI_BIAS, bias=bias)
        if train:
            self.dcos_w = torch.zeros(B, self.ldc, dtype=act_dtype, device=dev)
            self.dcos_x = torch.zeros(B, self.ldc, dtype=act_dtype, device=dev) if aam else self.dcos_w
            self.rowdot = torch.empty(B, dtype=f32, device=dev)
            # the two accumulators of a step (column dots of the AAM normalisation, d(emb) partial sums) share one
            # buffer so that ONE zero_ranges launch clears both
            self._zbuf = torch.empty(Cn + B * E, dtype=f32, device=dev)
            self._ztab = torch.tensor([[0, Cn + B * E]], dtype=torch.int64, device=dev)
            self.coldot = self._zbuf[:Cn]
            self.G1 = self._zbuf[Cn:].view(B, E)
            # per-element g * cos of the AAM normalisation backward; folded over the batch in a fixed order (colsum with
            # <= 128 rows has one writer per column): the class-weight gradient is bitwise reproducible
            self.colprod = torch.empty(B, Cn, dtype=f32, device=dev) if aam else None
            self.demb = torch.empty(B, E, dtype=f32, device=dev)
            # d emb = dcos_w [B, C] . W [C, E]: 66 rows = ONE row of tiles, so the class dimension is cut into S chunks
            # (one batched launch + the ragged rest) whose f32 partials are then summed in a fixed order -- the single
            # launch walked all 5994 classes on 12 workgroups (144 us)
            S = max(1, min(16, Cn // 256))
            Kc = (Cn // S) // 8 * 8 if S > 1 else Cn
            rest = Cn - S * Kc
            self.dx_parts = torch.empty(S + (1 if rest else 0), B, E, dtype=f32, device=dev)
            self.g_dx = [Gemm(B, E, Kc, self.dcos_w, w_operand, self.dx_parts, lda=self.ldc, ldb=E, ldc=E, transB=True,
                              batch=S, batch_inner=S, a_strides=(0, Kc), b_strides=(0, Kc * E), c_strides=(0, B * E))]
            if rest:
                self.g_dx.append(Gemm(B, E, rest, self.dcos_w.view(-1)[S * Kc:], w_operand.view(-1)[S * Kc * E:],
                                      self.dx_parts[S], lda=self.ldc, ldb=E, ldc=E, transB=True))
            # AAM class-weight gradient: product, column dots and the F.normalize backward in ONE launch (csrc/heads.hip
            # aam_dw_kernel) when E % 8 == 0; other embeddings keep the three-launch GEMM path
            self.fused_dw = aam and E % 8 == 0
            if aam and not self.fused_dw:
                self.H1 = torch.empty(Cn, E, dtype=f32, device=dev)
                self.g_dw = Gemm(Cn, E, B, self.dcos_x, self.emb_lp, self.H1, lda=self.ldc, ldb=E, ldc=E,
                                 transA=True, transB=True)
            elif not aam:
                self.g_dw = Gemm(Cn, E, B, self.dcos_x, self.emb_lp, w_grad, lda=self.ldc, ldb=E, ldc=E,
                                 transA=True, transB=True, accumulate=True)

    def forward_backward(self, label: torch.Tensor):
        """(loss, softmax[B,C]); in training also d(loss)/d(emb) -> self.demb and the head gradients
        (dW written for AAM, accumulated for CE; dbias accumulated) into the tensors given at build."""
        B, E, Cn = self.B, self.E, self.C
        assert label.dtype == torch.int64 and label.is_cuda and label.shape == (B,)
        aam, tr = self.kind == "aam", self.train
        if self.emb_lp is not self.emb:
            ops.cast(self.emb, self.emb_lp)
        if aam:
            ops.row_invnorm(self.emb, self.inv_x, B, E)
            ops.row_invnorm(self.w_master, self.inv_w, Cn, E)
        self.g_fwd()
        if tr:
            ops.zero_ranges(self._zbuf, self._ztab, blocks_per_range=16)
        ops.aam_softmax_fwd_bwd(self.logits, label, self.softmax, self.loss_rows,
                                self.dcos_w if tr else None, (self.dcos_x if aam else None) if tr else None,
                                self.inv_x if aam else None, self.inv_w if aam else None,
                                self.rowdot if (tr and aam) else None, self.colprod if (tr and aam) else None,
                                B, Cn, self.ldc, self.margin if aam else -1.0, self.scale, self.loss_scale if tr else None,
                                self.correct, easy_margin=self.easy_margin)
        loss = torch.empty((), dtype=torch.float32, device=self.loss_rows.device)     # (allocator only: no kernel)
        ops.mean(self.loss_rows, loss)
        if tr:
            for g in self.g_dx:
                g()
            ops.colsum(self.dx_parts, self.G1.view(-1), self.dx_parts.shape[0], B * E)    # <= 128 rows: one writer
            if aam:
                ops.normalize_bwd(self.G1, self.emb, self.inv_x, self.rowdot, self.demb, B, E)
                if self.fused_dw:
                    ops.aam_dw(self.dcos_x, self.emb_lp, self.colprod, self.w_master, self.inv_w, self.w_grad, B, Cn, E,
                               self.ldc)
                else:
                    self.g_dw()
                    ops.colsum(self.colprod, self.coldot, B, Cn)
                    ops.normalize_bwd(self.H1, self.w_master, self.inv_w, self.coldot, self.w_grad, Cn, E)
            else:
                self.demb.copy_(self.G1)
                self.g_dw()
                ops.colsum(self.dcos_w, self.bias_grad, B, Cn, self.ldc)
        return loss, self.softmax[:, :Cn]


class BceHead:
    """ref: wav2vec2_paired_input.py:200-206 + binary_cross_entropy.py:24-40: Linear(H -> 1) on the CLS token,
    BCE-with-logits (mean over pairs), prediction = sigmoid(logit).  Same interface as ClassifierHead."""

    def __init__(self, batch: int, embed_dim: int, *, w: torch.Tensor, b: torch.Tensor,
                 w_grad: Optional[torch.Tensor], b_grad: Optional[torch.Tensor], emb: torch.Tensor, train: bool,
                 loss_scale: Optional[torch.Tensor] = None):
        dev, f32 = emb.device, torch.float32
        self.loss_scale = loss_scale
        self.B, self.E, self.w, self.b, self.w_grad, self.b_grad, self.emb, self.train = (batch, embed_dim, w, b, w_grad,
                                                                                          b_grad, emb, train)
        self.prob = torch.empty(batch, dtype=f32, device=dev)
        self.loss_rows = torch.empty(batch, dtype=f32, device=dev)
        self.dlogit = torch.empty(batch, dtype=f32, device=dev) if train else None
        self.demb = torch.empty(batch, embed_dim, dtype=f32, device=dev) if train else None

    def forward_backward(self, label: torch.Tensor):
        """label [B] int64 in {0, 1} (1 = same speaker) -> (loss, prediction [B])."""
        assert label.dtype == torch.int64 and label.is_cuda and label.shape == (self.B,)
        tr = self.train
        ops.bce_head_fwd_bwd(self.emb, self.w.view(-1), self.b, label, self.prob, self.loss_rows,
                             self.dlogit if tr else None, self.demb if tr else None,
                             self.w_grad.view(-1) if tr else None, self.b_grad if tr else None, self.B, self.E,
                             self.loss_scale if tr else None)
        loss = torch.empty((), dtype=torch.float32, device=self.loss_rows.device)
        ops.mean(self.loss_rows, loss)
        return loss, self.prob



class FcStack:
    """Hidden ``nn.Sequential(nn.Linear, nn.ReLU)`` layers between the pooled embedding and the loss head
    (ref: src/lightning_modules/speaker/wav2vec2_fc.py:185-228 ``fc_list``, :363-412 pre/post speaker-embedding ops).
    f32 throughout (B x a few hundred features: negligible work, exact arithmetic) on the skinny linear kernels
    (csrc/skinny.hip): one launch per layer forward (bias + ReLU fused), two per layer backward (ReLU' fused)."""

    def __init__(self, store, batch: int, in_dim: int, hidden, x0: torch.Tensor, train: bool):
        dev, f32 = x0.device, torch.float32
        self.B, self.dims, self.train = batch, [in_dim] + list(hidden), train
        self.x = [x0] + [torch.empty(batch, h, dtype=f32, device=dev) for h in hidden]
        self.w = [store.p(f"fc_list.{i}.0.weight") for i in range(len(hidden))]
        self.b = [store.p(f"fc_list.{i}.0.bias") for i in range(len(hidden))]
        if train:
            self.dw = [store.g(f"fc_list.{i}.0.weight") for i in range(len(hidden))]
            self.db = [store.g(f"fc_list.{i}.0.bias") for i in range(len(hidden))]
            self.dx = [torch.empty(batch, d, dtype=f32, device=dev) for d in self.dims]      # d(loss)/d(x[i])

    def forward(self, upto: Optional[int] = None) -> torch.Tensor:
        """Run layers 0 .. upto (all by default); returns the output of the last one run."""
        n = len(self.w) if upto is None else upto + 1
        for i in range(n):
            ops.skinny_linear_fwd(self.x[i], self.w[i], self.b[i], self.x[i + 1], ops.ACT_RELU)
        return self.x[n]

    def backward(self, dout: torch.Tensor) -> torch.Tensor:
        """dout = d(loss)/d(output of the last layer) -> d(loss)/d(x0); parameter gradients accumulated."""
        n = len(self.w)
        self.dx[n].copy_(dout)
        for i in reversed(range(n)):
            ops.skinny_linear_bwd_w(self.dx[i + 1], self.x[i + 1], self.x[i], self.dw[i], self.db[i], ops.ACT_RELU, True)
            ops.skinny_linear_bwd_x(self.dx[i + 1], self.x[i + 1], self.w[i], self.dx[i], ops.ACT_RELU)
        return self.dx[0]
