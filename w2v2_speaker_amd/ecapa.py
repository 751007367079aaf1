"""ECAPA-TDNN training step on the HIP path (BASELINE configs[4]).

ref: src/lightning_modules/speaker/ecapa_tdnn.py:51-137 (``EcapaTdnnModule`` -> speechbrain 0.5.x
``ECAPA_TDNN(input_size=40, channels=[1024]*4+[3072], kernel_sizes=[5,3,3,3,1], dilations=[1,2,3,4,1],
attention_channels=128, res2net_scale=8, se_channels=128, lin_neurons=192)``, config/network/ecapa_tdnn.yaml) with
the AAM-softmax head on the 192-d embedding (the ``Classifier`` is skipped under AAM, :98-100,129-137).
speechbrain is not part of the reference tree: the arithmetic is the published definition restated in
``oracle/ecapa_oracle.py`` -- parity for this row is against that restatement only ("parity unpinned").

Same design as the wav2vec2 engine: channels-last activations [B*T, C], a flat f32 parameter arena (+ bf16 operand
copy) with fused Adam, static buffers and prebuilt GEMM descriptors, hand-written backward.  Dilated "same"
convolutions are reflect-im2col + GEMM (csrc/tdnn.hip); Res2Net slices and the MFA concatenation are row-strided
views of their parent tensors; BatchNorm uses deterministic two-stage batch statistics.
"""
from __future__ import annotations

import os

import dataclasses
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .asp import AttentivePool
from .heads import ClassifierHead
from .ops import EPI_ADD, EPI_BIAS, EPI_NONE, Gemm

FE = "feature_extractor."          # reference attribute name of the ECAPA_TDNN module (ecapa_tdnn.py:75)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
ALIGN = 64


@dataclasses.dataclass
class EcapaConfig:
    """ref: config/network/ecapa_tdnn.yaml:4-30 (EcapaTDNNModuleConfig)."""
    input_mel_coefficients: int = 40
    lin_neurons: int = 192
    channels: Tuple[int, ...] = (1024, 1024, 1024, 1024, 3072)
    kernel_sizes: Tuple[int, ...] = (5, 3, 3, 3, 1)
    dilations: Tuple[int, ...] = (1, 2, 3, 4, 1)
    attention_channels: int = 128
    res2net_scale: int = 8
    se_channels: int = 128

    @staticmethod
    def tiny() -> "EcapaConfig":
        return EcapaConfig(input_mel_coefficients=16, lin_neurons=24, channels=(64, 64, 64, 64, 192),
                           attention_channels=16, res2net_scale=4, se_channels=16)


def ecapa_param_shapes(cfg: EcapaConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """speechbrain state-dict names under ``feature_extractor.`` (Conv1d wrapper -> .conv, BatchNorm1d -> .norm)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def tdnn(p, cin, cout, k):
        s[FE + p + "conv.conv.weight"] = (cout, cin, k)
        s[FE + p + "conv.conv.bias"] = (cout,)
        s[FE + p + "norm.norm.weight"] = (cout,)
        s[FE + p + "norm.norm.bias"] = (cout,)
    C = cfg.channels
    tdnn("blocks.0.", cfg.input_mel_coefficients, C[0], cfg.kernel_sizes[0])
    for i in range(1, len(C) - 1):
        p = f"blocks.{i}."
        assert C[i - 1] == C[i], "SERes2NetBlock shortcut conv (in != out channels) is not used by the reference config"
        tdnn(p + "tdnn1.", C[i - 1], C[i], 1)
        w = C[i] // cfg.res2net_scale
        for j in range(cfg.res2net_scale - 1):
            tdnn(p + f"res2net_block.blocks.{j}.", w, w, cfg.kernel_sizes[i])
        tdnn(p + "tdnn2.", C[i], C[i], 1)
        s[FE + p + "se_block.conv1.conv.weight"] = (cfg.se_channels, C[i], 1)
        s[FE + p + "se_block.conv1.conv.bias"] = (cfg.se_channels,)
        s[FE + p + "se_block.conv2.conv.weight"] = (C[i], cfg.se_channels, 1)
        s[FE + p + "se_block.conv2.conv.bias"] = (C[i],)
    tdnn("mfa.", C[-1], C[-1], cfg.kernel_sizes[-1])
    A = cfg.attention_channels
    s[FE + "asp.tdnn.conv.conv.weight"] = (A, 3 * C[-1], 1)
    s[FE + "asp.tdnn.conv.conv.bias"] = (A,)
    s[FE + "asp.tdnn.norm.norm.weight"] = (A,)
    s[FE + "asp.tdnn.norm.norm.bias"] = (A,)
    s[FE + "asp.conv.conv.weight"] = (C[-1], A, 1)
    s[FE + "asp.conv.conv.bias"] = (C[-1],)
    s[FE + "asp_bn.norm.weight"] = (2 * C[-1],)
    s[FE + "asp_bn.norm.bias"] = (2 * C[-1],)
    s[FE + "fc.conv.weight"] = (cfg.lin_neurons, 2 * C[-1], 1)
    s[FE + "fc.conv.bias"] = (cfg.lin_neurons,)
    return s


class EcapaStore:
    """Flat parameter arena (f32 master, gradient, Adam moments, bf16 operand copy) of the ECAPA model + AAM head;
    the subset of ParamStore's interface that AttentivePool / ClassifierHead / the plan below use."""

    def __init__(self, cfg: EcapaConfig, device, act_dtype: torch.dtype = torch.bfloat16, num_speakers: int = 5994):
        assert act_dtype in (torch.bfloat16, torch.float32), "ECAPA: bf16 or f32 (the reference runs it in fp32; no loss scaler here)"
        self.cfg, self.device, self.act_dtype, self.num_speakers = cfg, torch.device(device), act_dtype, num_speakers
        self.embed_dim = cfg.lin_neurons
        shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        shapes["loss_fn.fc_weights"] = (num_speakers, cfg.lin_neurons)
        shapes.update(ecapa_param_shapes(cfg))
        self.shapes, self.offsets = shapes, {}
        off = 0
        for n, s in shapes.items():
            self.offsets[n] = off
            off += (int(np.prod(s)) + ALIGN - 1) // ALIGN * ALIGN
        self.n_total = self.n_train = off
        dev = self.device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = self.exp_avg_sq = None
        self.flat_lp = torch.zeros(off, dtype=act_dtype, device=dev) if ops.is16(act_dtype) else None
        # BatchNorm1d buffers of EVERY norm layer live in the store (shared by the training and the evaluation plans,
        # saved / loaded under the speechbrain names ``...norm.running_mean`` / ``running_var``): one f32 record
        # {running_mean[C], running_var[C]} per layer, keyed by the name of its ``.weight``
        self.bn_running: Dict[str, torch.Tensor] = {}
        for n, shp in shapes.items():
            if n.endswith("norm.weight"):
                self.bn_running[n] = torch.cat([torch.zeros(shp[0]), torch.ones(shp[0])]).to(dev)
        self.asp_running = self.bn_running[FE + "asp.tdnn.norm.norm.weight"]
        self.bn_batches_tracked = 0
        self.version, self.step_count = 0, 0

    def running(self, weight_name: str) -> torch.Tensor:
        """{running_mean, running_var} record of the BatchNorm whose scale parameter is ``weight_name``."""
        return self.bn_running[weight_name]

    def _buffer_views(self) -> "OrderedDict[str, torch.Tensor]":
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for n, r in self.bn_running.items():
            C = r.numel() // 2
            base = n[:-len("weight")]
            out[base + "running_mean"], out[base + "running_var"] = r[:C], r[C:]
        return out

    def _view(self, buf, name):
        s, o = self.shapes[name], self.offsets[name]
        return buf[o:o + int(np.prod(s))].view(*s)

    def p(self, name): return self._view(self.flat, name)
    def g(self, name): return self._view(self.grad, name)
    def w(self, name): return self._view(self.flat_lp if self.flat_lp is not None else self.flat, name)

    def sync_lowp(self) -> None:
        if self.flat_lp is not None:
            ops.cast(self.flat, self.flat_lp)
        self.version += 1

    def zero_grad(self) -> None:
        if self.grad.is_cuda:
            if getattr(self, "_ztab", None) is None:
                self._ztab = torch.tensor([[0, self.grad.numel()]], dtype=torch.int64, device=self.device)
            ops.zero_ranges(self.grad, self._ztab, blocks_per_range=1024)
        else:
            self.grad.zero_()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        seen = set()
        bufs = self._buffer_views()
        for k, v in sd.items():
            bk = k if k in bufs else (FE + k if FE + k in bufs else None)
            if bk is not None:                             # BatchNorm running statistics
                bufs[bk].copy_(torch.as_tensor(v).to(self.device, torch.float32))
                continue
            if k.endswith("num_batches_tracked"):
                self.bn_batches_tracked = int(v)
                continue
            name = k if k in self.shapes else (FE + k if FE + k in self.shapes else None)
            if name is None:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            t = torch.as_tensor(v).to(torch.float32)
            if tuple(t.shape) != tuple(self.shapes[name]):
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {self.shapes[name]}")
            self.p(name).copy_(t.to(self.device))
            seen.add(name)
        if strict and len(seen) != len(self.shapes):
            raise KeyError(f"missing keys: {[n for n in self.shapes if n not in seen][:5]}")
        self.sync_lowp()

    def state_dict(self):
        sd = OrderedDict((n, self.p(n).detach().clone().cpu()) for n in self.shapes)
        for n, b in self._buffer_views().items():
            sd[n] = b.detach().clone().cpu()
            if n.endswith("running_var"):
                sd[n[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(self.bn_batches_tracked)
        return sd

    def init_weights(self, seed: int = 20211) -> None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n, s in self.shapes.items():
            if n.endswith("norm.weight"):
                t = torch.ones(s)
            elif n.endswith("bias"):
                t = torch.zeros(s)
            elif n == "loss_fn.fc_weights":
                t = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[0] + s[1]))
            else:                                    # torch Conv1d default: kaiming-uniform, bound 1/sqrt(fan_in)
                t = (torch.rand(s, generator=g) * 2 - 1) / math.sqrt(s[1] * s[2])
            self.p(n).copy_(t.to(self.device))
        self.sync_lowp()

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                  grad_scale: float = 1.0) -> None:
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.grad), torch.zeros_like(self.grad)
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.flat_lp, self.n_total, lr, beta1, beta2,
                      eps, self.step_count, grad_scale)
        self.version += 1


def _cols(v: torch.Tensor, c0: int, c1: Optional[int] = None) -> torch.Tensor:
    """Column slice of a row-padded activation buffer that keeps its zero-padded parent (``_full``)."""
    out = v[:, c0:c1]
    if hasattr(v, "_full"):
        out._full = v._full[:, c0:c1]
    return out



def f32_dw_split(n_out: int, n_in: int, tokens: int, slots: int = 512) -> int:
    """Split-K factor of an exact-f32 weight-gradient product dW[n_out, n_in] = dY^T X over `tokens` rows (f32 atomics
    into a zeroed target).  The output is only n_out x n_in / 128^2 tiles -- 3 for a Res2Net chunk, 576 for the last
    3072 x 3072 convolution -- on 512 workgroup slots (2 per CU).  Few tiles: fill the slots, at most 32 ways (more
    splits of a 3-tile output lose to the contention of their atomics: 77 ways measured 139 us against 115 at 32).
    More tiles than half the slots: the factor that minimises rounds / factor, i.e. the launch's quantisation loss
    (576 tiles unsplit = 2 rounds, the second 12 % full: 5.83 ms; 8 ways = 9 full rounds of an eighth: 4.31 ms).
    At least 256 tokens per split."""
    tiles = -(-n_out // 128) * -(-n_in // 128)
    if 2 * tiles <= slots:
        sk = max(1, min(32, slots // tiles))
    else:
        sk, best = 1, None
        for c in range(1, 17):
            cost = -(-tiles * c // slots) / c * (1.0 + 0.004 * c)
            if best is None or cost < best - 1e-9:
                sk, best = c, cost
    while sk > 1 and tokens // sk < 256:
        sk -= 1
    return sk


class _Tdnn:
    """TDNNBlock = Conv1d("same", reflect, dilation) -> ReLU -> BatchNorm1d over one [M, Cin] view -> [M, Cout] view."""

    def __init__(self, plan: "EcapaPlan", prefix: str, x: torch.Tensor, ldx: int, cin: int, cout: int, k: int, dil: int,
                 y: torch.Tensor, ldy: int, x2: Optional[torch.Tensor] = None, ldx2: int = 0,
                 shared: Optional[dict] = None):
        """x2 (k > 1 only): the block convolves x + x2 -- the im2col pass sums the two while it gathers the taps.
        shared (exact-f32 training, Res2Net chunks): {"col", "da", "dwp"} views into the owner's [chunks, ...] arrays -- the
        owner then computes the weight gradients of all its chunks in ONE batched product (_SERes2Net.backward)."""
        st, B, T, dev, adt, f32 = plan.store, plan.B, plan.T, plan.dev, plan.adt, torch.float32
        assert x2 is None or k > 1, "a second input operand needs the im2col pass (k > 1)"
        self.plan, self.pre, self.x, self.ldx, self.y, self.ldy = plan, FE + prefix, x, ldx, y, ldy
        self.x2, self.ldx2 = x2, ldx2
        self.cin, self.cout, self.k, self.dil = cin, cout, k, dil
        M, K = B * T, k * cin
        self.M, self.K = M, K
        self.wp, self.wpt = plan.weight_pair(cout, K)                          # packed [cout][tap][cin] operand (+ its transpose)
        self.shared = shared
        self.col = (shared["col"] if shared else plan.buf(M, K)) if k > 1 else None
        self.a = torch.empty(M, cout, dtype=adt, device=dev)                   # pre-activation (saved)
        self.mean_rstd = torch.empty(cout, 2, dtype=f32, device=dev)
        self.running = st.running(self.pre + "norm.norm.weight")            # shared BatchNorm1d buffers (store)
        self.work = ops.bn_workspace(M, cout, dev)
        A, lda = (self.col, K) if k > 1 else (x, ldx)
        self.g_fwd = Gemm(M, cout, K, A, self.wp, self.a, lda=lda, ldb=K, ldc=cout, epilogue=EPI_BIAS,
                          bias=st.p(self.pre + "conv.conv.bias"))
        if plan.train:
            self.da = shared["da"] if shared else plan.buf(M, cout)
            self.cs_part = torch.empty(ops.bn_colsum_rows(M, cout), cout, dtype=f32, device=dev)   # bias-gradient partials
            self.dwp = (shared["dwp"] if shared else torch.zeros(cout, K, dtype=f32, device=dev)) if k > 1 else None
            dW = self.dwp if k > 1 else st.g(self.pre + "conv.conv.weight").view(cout, cin)
            # bf16: weight + bias gradient through the grouped, atomic-free wgrad kernels (K-major operands: da and the
            # conv input / im2col buffer, both zero-padded to a multiple of 64 rows), launched per group by the owner
            self.grouped = ops.is16(adt) and hasattr(A, "_full")
            if self.grouped:
                self.wg_problem = (self.da._full, A._full, dW, st.g(self.pre + "conv.conv.bias"))
            # exact-f32 mode (no grouped launch): K of this product = the B * T tokens, its output only
            # cout x K / 128^2 tiles (3 for a Res2Net chunk) -- split the token dimension over ~2 workgroups per CU
            # (f32 atomics into a zeroed target), like the wav2vec2 engine's f32 weight gradients
            self._dw_split = f32_dw_split(cout, K, M) if not self.grouped else 1
            self._dwp_ztab = (torch.tensor([[0, cout * K]], dtype=torch.int64, device=dev)
                              if (k > 1 and self._dw_split > 1) else None)
            self.g_dw = Gemm(cout, K, M, self.da, A, dW, lda=cout, ldb=lda, ldc=K, transA=True, transB=True,
                             split_k=self._dw_split, accumulate=(k == 1 or self._dw_split > 1))
            self.dcol = torch.empty(M, K, dtype=adt, device=dev) if k > 1 else None
            # bf16: the data-gradient product reads the packed weight TRANSPOSED ([K][cout], refreshed with the pack)
            # so that it is an NT product on the LDS-DMA ring kernels instead of a K-major-B one on the generic kernel
            # (self.wpt: [K][cout] view of the plan's transposed-weight arena, refreshed by ONE batched-transpose launch
            # for all TDNN blocks -- EcapaPlan._refresh)
            self._dx_gemm = {}

    def refresh(self) -> None:
        ops.pack_conv_weight(self.plan.store.p(self.pre + "conv.conv.weight"), self.wp)

    def forward(self) -> None:
        st, pl = self.plan.store, self.plan
        if self.k > 1:
            ops.im2col_reflect(self.x, self.ldx, self.col, pl.B, pl.T, self.cin, self.k, self.dil, self.x2, self.ldx2)
        self.g_fwd()
        ops.bn_fwd(self.a, self.cout, self.work, self.mean_rstd, self.running, st.p(self.pre + "norm.norm.weight"),
                   st.p(self.pre + "norm.norm.bias"), self.y, self.ldy, self.M, self.cout, BN_EPS, BN_MOMENTUM, True,
                   pl.train)

    def weight_grad_single(self) -> None:
        """dW, db of this block alone (f32 mode, or a block that is not part of a deferred group)."""
        st = self.plan.store
        if self.grouped:
            if not hasattr(self, "_wg"):
                self._wg = ops.WgradGroup([self.wg_problem], self.M, self.da._full.shape[0])
            self._wg()
        else:
            # (the BatchNorm backward left the column sums of da per row block: 78 rows to fold instead of 19800)
            ops.colsum(self.cs_part, st.g(self.pre + "conv.conv.bias"), self.cs_part.shape[0], self.cout)
            if self._dwp_ztab is not None:
                ops.zero_ranges(self.dwp.view(-1), self._dwp_ztab, blocks_per_range=64)
            self.g_dw()
        self.finish_weight_grad()

    def finish_weight_grad(self) -> None:
        if self.k > 1:       # packed [cout][tap][cin] gradient -> torch layout [cout][cin][tap] (added into the arena)
            ops.unpack_conv_grad(self.dwp.view(self.cout, self.k, self.cin),
                                 self.plan.store.g(self.pre + "conv.conv.weight"))

    def backward(self, dy: torch.Tensor, lddy: int, dx: Optional[torch.Tensor], lddx: int, accumulate: bool,
                 defer_dw: bool = False, dy2: Optional[torch.Tensor] = None, lddy2: int = 0) -> None:
        """dy = gradient of the block output (row stride lddy); dx (None: input needs no gradient) receives or, with
        ``accumulate``, is incremented by the input gradient.  Parameter gradients go to the arena; with ``defer_dw``
        (bf16) the owner launches the weight gradients of several blocks in one grouped call afterwards."""
        st, pl = self.plan.store, self.plan
        ops.bn_bwd(dy, lddy, self.a, self.cout, self.mean_rstd, st.p(self.pre + "norm.norm.weight"), self.work,
                   st.g(self.pre + "norm.norm.weight"), st.g(self.pre + "norm.norm.bias"), self.da, self.cout, self.M,
                   self.cout, True, colsum_partial=None if self.grouped else self.cs_part, dy2=dy2, lddy2=lddy2)
        if not ((self.grouped or self.shared) and defer_dw):
            self.weight_grad_single()
        if dx is None:
            return
        key = (dx.data_ptr(), lddx, accumulate)
        if key not in self._dx_gemm:
            W, ldb, tb = (self.wpt, self.cout, False) if self.wpt is not None else (self.wp, self.K, True)
            if self.k > 1:
                self._dx_gemm[key] = Gemm(self.M, self.K, self.cout, self.da, W, self.dcol, lda=self.cout, ldb=ldb,
                                          ldc=self.K, transB=tb)
            elif accumulate:
                self._dx_gemm[key] = Gemm(self.M, self.K, self.cout, self.da, W, dx, lda=self.cout, ldb=ldb,
                                          ldc=lddx, transB=tb, epilogue=EPI_ADD, aux=dx, ldaux=lddx)
            else:
                self._dx_gemm[key] = Gemm(self.M, self.K, self.cout, self.da, W, dx, lda=self.cout, ldb=ldb,
                                          ldc=lddx, transB=tb)
        self._dx_gemm[key]()
        if self.k > 1:
            ops.col2im_reflect(self.dcol, dx, lddx, pl.B, pl.T, self.cin, self.k, self.dil, accumulate)


class _SEBlock:
    """s = mean_t x -> relu(W1 s + b1) -> sigmoid(W2 . + b2) = gate g [B, C];  y = x * g."""

    def __init__(self, plan: "EcapaPlan", prefix: str, x: torch.Tensor, y: torch.Tensor, C: int):
        st, B, dev, f32 = plan.store, plan.B, plan.dev, torch.float32
        S = plan.cfg.se_channels
        self.plan, self.pre, self.x, self.y, self.C, self.S = plan, FE + prefix, x, y, C, S
        self.s = torch.empty(B, C, dtype=f32, device=dev)
        self.z = torch.empty(B, S, dtype=f32, device=dev)
        self.g = torch.empty(B, C, dtype=f32, device=dev)
        p = st.p
        self.W1, self.W2 = p(self.pre + "conv1.conv.weight").view(S, C), p(self.pre + "conv2.conv.weight").view(C, S)
        self.b1, self.b2 = p(self.pre + "conv1.conv.bias"), p(self.pre + "conv2.conv.bias")
        if plan.train:
            self.dg = torch.empty(B, C, dtype=f32, device=dev)
            self.dz = torch.empty(B, S, dtype=f32, device=dev)
            self.ds = torch.empty(B, C, dtype=f32, device=dev)

    def forward(self) -> None:
        pl = self.plan
        ops.pool_fwd(self.x.view(pl.B, pl.T, self.C), self.s, ops.POOL_MODES["mean"])
        ops.skinny_linear_fwd(self.s, self.W1, self.b1, self.z, ops.ACT_RELU)
        ops.skinny_linear_fwd(self.z, self.W2, self.b2, self.g, ops.ACT_SIGMOID)
        ops.se_scale(self.x, self.g, self.y, pl.B, pl.T, self.C)

    def backward(self, dy: torch.Tensor, dx: torch.Tensor) -> None:
        """dy [M, C] contiguous -> dx [M, C] contiguous (written)."""
        pl, g = self.plan, self.plan.store.g
        ops.se_bwd_gate(dy, self.x, self.dg, pl.B, pl.T, self.C)                  # d/d gate (before sigmoid')
        ops.skinny_linear_bwd_w(self.dg, self.g, self.z, g(self.pre + "conv2.conv.weight").view(self.C, self.S),
                                g(self.pre + "conv2.conv.bias"), ops.ACT_SIGMOID, True)
        ops.skinny_linear_bwd_x(self.dg, self.g, self.W2, self.dz, ops.ACT_SIGMOID)          # d/d z (before relu')
        ops.skinny_linear_bwd_w(self.dz, self.z, self.s, g(self.pre + "conv1.conv.weight").view(self.S, self.C),
                                g(self.pre + "conv1.conv.bias"), ops.ACT_RELU, True)
        ops.skinny_linear_bwd_x(self.dz, self.z, self.W1, self.ds, ops.ACT_RELU)
        ops.se_bwd_x(dy, self.g, self.ds, dx, pl.B, pl.T, self.C)


class _SERes2Net:
    """tdnn1 (1x1) -> Res2Net (cumulative dilated k=3 TDNNs on channel slices) -> tdnn2 (1x1) -> SE -> + input."""

    def __init__(self, plan: "EcapaPlan", idx: int, x: torch.Tensor, ldx: int, out: torch.Tensor, ldo: int):
        cfg, B, T, dev, adt = plan.cfg, plan.B, plan.T, plan.dev, plan.adt
        C, sc = cfg.channels[idx], cfg.res2net_scale
        w, M = C // sc, B * T
        p = f"blocks.{idx}."
        self.plan, self.C, self.w, self.sc, self.x, self.ldx, self.out, self.ldo, self.M = plan, C, w, sc, x, ldx, out, ldo, M
        e = lambda c: plan.buf(M, c)
        self.t1, self.r2, self.t2, self.se_out = e(C), e(C), e(C), e(C)
        self.tdnn1 = _Tdnn(plan, p + "tdnn1.", x, ldx, C, C, 1, 1, self.t1, C)
        # chunk i >= 2 convolves x_i + y_{i-1}: with k > 1 the im2col pass sums the two slices while it gathers the taps
        # (no sum tensor, no add launch); k = 1 keeps a materialised sum
        fuse = cfg.kernel_sizes[idx] > 1
        self.sums = [None, None] + [None if fuse else e(w) for _ in range(2, sc)]
        self.chunks: List[Optional[_Tdnn]] = [None]
        # exact-f32 training: the sc - 1 chunk convolutions share one [chunks, M, .] array per operand, so that their weight
        # gradients -- 3-tile outputs over 19800 tokens, 30 us each as separate split-K launches -- are ONE batched product
        kk = cfg.kernel_sizes[idx]
        self.batched = bool(plan.train and not ops.is16(adt) and kk > 1 and sc > 2 and
                            not os.environ.get("W2V2_ECAPA_NO_BATCHED_DW"))
        if self.batched:
            n, Mp, Kc = sc - 1, (M + 63) // 64 * 64, kk * w
            self._col_all = torch.zeros(n, Mp, Kc, dtype=adt, device=dev)
            self._da_all = torch.zeros(n, Mp, w, dtype=adt, device=dev)
            self._dwp_all = torch.zeros(n, w, Kc, dtype=torch.float32, device=dev)
        for i in range(1, sc):
            x2, ldx2 = (self.r2[:, (i - 1) * w:i * w], C) if (fuse and i >= 2) else (None, 0)
            src, ld = (_cols(self.t1, i * w, (i + 1) * w), C) if (i == 1 or fuse) else (self.sums[i], w)
            shared = ({"col": self._col_all[i - 1, :M], "da": self._da_all[i - 1, :M], "dwp": self._dwp_all[i - 1]}
                      if self.batched else None)
            self.chunks.append(_Tdnn(plan, p + f"res2net_block.blocks.{i - 1}.", src, ld, w, w,
                                     cfg.kernel_sizes[idx], cfg.dilations[idx], self.r2[:, i * w:(i + 1) * w], C,
                                     x2=x2, ldx2=ldx2, shared=shared))
        if self.batched:
            n, Mp, Kc = sc - 1, (M + 63) // 64 * 64, kk * w
            # split-K so that the batch fills the 512 workgroup slots once (7 chunks x 3 tiles x 24: measured best of 4 .. 32,
            # 23.66 -> 23.3 ms/step; W2V2_ECAPA_BDW_SPLIT overrides)
            tiles = -(-w // 128) * -(-Kc // 128)
            split = int(os.environ.get("W2V2_ECAPA_BDW_SPLIT", "0")) or max(1, min(32, 512 // (n * tiles)))
            self._dw_ztab = torch.tensor([[0, n * w * Kc]], dtype=torch.int64, device=dev)
            self._g_dw_all = Gemm(w, Kc, M, self._da_all, self._col_all, self._dwp_all, lda=w, ldb=Kc, ldc=Kc, transA=True,
                                  transB=True, batch=n, a_strides=(Mp * w, 0), b_strides=(Mp * Kc, 0),
                                  c_strides=(w * Kc, 0), split_k=split, accumulate=True)
        self.tdnn2 = _Tdnn(plan, p + "tdnn2.", self.r2, C, C, C, 1, 1, self.t2, C)
        self.se = _SEBlock(plan, p + "se_block.", self.t2, self.se_out, C)
        if plan.train:
            self.d_se, self.d_t2, self.d_r2, self.d_t1 = e(C), e(C), e(C), e(C)

    def blocks(self):
        return [self.tdnn1] + [c for c in self.chunks if c is not None] + [self.tdnn2]

    def forward(self) -> None:
        w, C, M = self.w, self.C, self.M
        self.tdnn1.forward()
        ops.copy_strided(self.t1, C, self.r2, C, M, w)                             # chunk 0 passes through
        for i in range(1, self.sc):
            if i >= 2 and self.sums[i] is not None:
                ops.add_strided(self.t1[:, i * w:], C, self.r2[:, (i - 1) * w:], C, self.sums[i], w, M, w)
            self.chunks[i].forward()
        self.tdnn2.forward()
        self.se.forward()
        ops.add_strided(self.se_out, C, self.x, self.ldx, self.out, self.ldo, M, C)

    def backward(self, dout: torch.Tensor, lddo: int, dx: torch.Tensor, lddx: int, accumulate: bool) -> None:
        """dout = gradient of the block output (view, row stride lddo); dx (+)= gradient of the block input."""
        w, C, M, sc = self.w, self.C, self.M, self.sc
        # residual branch: d(input) gets dout; SE branch: dout -> d_t2
        ops.copy_strided(dout, lddo, self.d_se, C, M, C)
        self.se.backward(self.d_se, self.d_t2)
        self.tdnn2.backward(self.d_t2, C, self.d_r2, C, False, defer_dw=True)
        # Res2Net, last slice first: the gradient of (x_i + y_{i-1}) lands in d_t1[:, i] and is carried to y_{i-1}
        for i in range(sc - 1, 0, -1):
            # chunk i < sc - 1 also feeds chunk i + 1: its output gradient is d_r2[:, i] + d_t1[:, i + 1], which the
            # BatchNorm backward sums as it reads (no sum tensor, no add launch)
            dy2, ld2 = (self.d_t1[:, (i + 1) * w:(i + 2) * w], C) if i < sc - 1 else (None, 0)
            self.chunks[i].backward(self.d_r2[:, i * w:(i + 1) * w], C, self.d_t1[:, i * w:(i + 1) * w], C, False,
                                    defer_dw=True, dy2=dy2, lddy2=ld2)
        if self.batched:            # the chunks' weight gradients: one zeroing, one batched split-K product, then per chunk the
            st = self.plan.store    # bias fold (BatchNorm backward's row-block partials) and the unpack into the arena
            ops.zero_ranges(self._dwp_all.view(-1), self._dw_ztab, blocks_per_range=64)
            self._g_dw_all()
            for c in self.chunks[1:]:
                ops.colsum(c.cs_part, st.g(c.pre + "conv.conv.bias"), c.cs_part.shape[0], c.cout)
                c.finish_weight_grad()
        ops.copy_strided(self.d_r2, C, self.d_t1, C, M, w)
        self.tdnn1.backward(self.d_t1, C, dx, lddx, accumulate, defer_dw=True)
        # residual: dx += dout
        ops.add_strided(dx, lddx, dout, lddo, dx, lddx, M, C)
        # the weight gradients of the block (tdnn1, tdnn2, the Res2Net TDNNs) are independent of the chain above: 16-bit
        # mode leaves them to ONE grouped launch at the end of the plan's backward (EcapaPlan.backward)


class EcapaPlan:
    """Static plan of one ECAPA-TDNN forward (+ backward) for a fixed [B, T, n_mels] input."""

    def __init__(self, store: EcapaStore, batch: int, frames: int, *, train: bool, aam_margin: float = 0.2,
                 aam_scale: float = 30.0):
        cfg = store.cfg
        self.store, self.cfg, self.B, self.T, self.train = store, cfg, batch, frames, train
        self.dev, self.adt = store.device, store.act_dtype
        B, T, M, dev, adt, f32 = batch, frames, batch * frames, self.dev, self.adt, torch.float32
        C = cfg.channels
        nb = len(C) - 2                                                       # SE-Res2Net blocks
        assert all(c == C[1] for c in C[1:-1]) and C[-1] == nb * C[1], "MFA concatenates the SE-Res2Net outputs"
        # packed conv weights of all TDNN blocks in one arena (+ a second one with their transposes for the 16-bit
        # data-gradient products): the transposes are refreshed by one w2v2_transpose_many launch per optimiser step
        self._wp_pool = torch.empty(store.n_total, dtype=adt, device=dev)
        self._wpt_pool = torch.empty(store.n_total, dtype=adt, device=dev) if (train and ops.is16(adt)) else None
        self._wp_off, self._t_rows = 0, []
        F_ = cfg.input_mel_coefficients
        self.feat = self.buf(M, F_)
        self.x0 = self.buf(M, C[0])
        self.cat = self.buf(M, C[-1])                                         # MFA input: block outputs side by side
        self.block0 = _Tdnn(self, "blocks.0.", self.feat, F_, F_, C[0], cfg.kernel_sizes[0], cfg.dilations[0],
                            self.x0, C[0])
        self.blocks: List[_SERes2Net] = []
        for i in range(1, nb + 1):
            x, ldx = (self.x0, C[0]) if i == 1 else (_cols(self.cat, (i - 2) * C[1], (i - 1) * C[1]), C[-1])
            self.blocks.append(_SERes2Net(self, i, x, ldx, self.cat[:, (i - 1) * C[1]:], C[-1]))
        rows = (M + 63) // 64 * 64
        self.mfa_out = self.buf(M, C[-1])
        self.mfa_out._w2v2_padded = self.mfa_out._full
        self.mfa = _Tdnn(self, "mfa.", self.cat, C[-1], C[-1], C[-1], cfg.kernel_sizes[-1], cfg.dilations[-1],
                         self.mfa_out, C[-1])
        self.pooled = torch.empty(B, 2 * C[-1], dtype=f32, device=dev)
        self.d_mfa = torch.zeros(rows, C[-1], dtype=adt, device=dev)[:M] if train else None
        self.asp = AttentivePool(store, self.mfa_out, self.pooled, self.d_mfa, B, T, train, prefix=FE + "asp.")
        E2, L = 2 * C[-1], cfg.lin_neurons
        self.bn_mr = torch.empty(E2, 2, dtype=f32, device=dev)
        self.bn_running = store.running(FE + "asp_bn.norm.weight")
        self.bn_work = ops.bn_workspace(B, E2, dev)
        self.e2 = torch.empty(B, E2, dtype=f32, device=dev)
        self.emb = torch.empty(B, L, dtype=f32, device=dev)
        p, g = store.p, store.g
        Wfc = p(FE + "fc.conv.weight").view(L, E2)
        self.Wfc = Wfc
        self.head = ClassifierHead("aam", B, L, store.num_speakers, w_master=p("loss_fn.fc_weights"),
                                   w_operand=store.w("loss_fn.fc_weights"),
                                   w_grad=g("loss_fn.fc_weights") if train else None, emb=self.emb, act_dtype=adt,
                                   train=train, margin=aam_margin, scale=aam_scale)
        if train:
            self.de2 = torch.empty(B, E2, dtype=f32, device=dev)
            self.dpooled = torch.empty(B, E2, dtype=f32, device=dev)
            self.d_cat = torch.empty(M, C[-1], dtype=adt, device=dev)
            self.d_x0 = torch.empty(M, C[0], dtype=adt, device=dev)
        self._version = -1
        self._wgs = None

    def weight_pair(self, cout: int, K: int):
        """([cout, K] packed-weight view, [K, cout] transposed view or None) out of the plan's two weight arenas."""
        o, n = self._wp_off, cout * K
        self._wp_off = o + (n + ALIGN - 1) // ALIGN * ALIGN
        assert self._wp_off <= self._wp_pool.numel()
        wp = self._wp_pool[o:o + n].view(cout, K)
        wpt = None
        if self._wpt_pool is not None:
            wpt = self._wpt_pool[o:o + n].view(K, cout)
            self._t_rows.append([o, o, cout, K])
        return wp, wpt

    def buf(self, rows: int, cols: int) -> torch.Tensor:
        """[rows, cols] activation whose storage is zero-padded to a multiple of 64 rows (``_full``): legal K-major
        operand of the grouped weight-gradient kernels (csrc/wgrad.hip contract)."""
        full = torch.zeros((rows + 63) // 64 * 64, cols, dtype=self.adt, device=self.dev)
        v = full[:rows]
        v._full = full
        return v

    def _tdnns(self) -> List[_Tdnn]:
        out = [self.block0]
        for b in self.blocks:
            out += b.blocks()
        return out + [self.mfa]

    def _refresh(self) -> None:
        if self._version != self.store.version:
            for t in self._tdnns():
                t.refresh()
            if self._wpt_pool is not None and self._t_rows:
                if getattr(self, "_t_table", None) is None:
                    self._t_table = torch.tensor(self._t_rows, dtype=torch.int64, device=self.dev)
                ops.transpose_many(self._wp_pool, self._wpt_pool, self._t_table, self._t_table.shape[0])
            self._version = self.store.version

    def embed(self, feat: torch.Tensor) -> torch.Tensor:
        """ref: ecapa_tdnn.py:110-118 (compute_speaker_embedding): feat [B, T, n_mels] -> [B, lin_neurons] f32."""
        st, B, T = self.store, self.B, self.T
        assert feat.shape == (B, T, self.cfg.input_mel_coefficients) and feat.is_cuda
        self._refresh()
        f2 = feat.reshape(B * T, -1)
        if self.feat.dtype == torch.float32:        # (input plumbing: the filterbank frames into the plan's padded buffer)
            ops.copy_strided(f2, f2.shape[1], self.feat, self.feat.stride(0), B * T, f2.shape[1])
        else:
            ops.cast(f2.contiguous(), self.feat)
        self.block0.forward()
        for b in self.blocks:
            b.forward()
        self.mfa.forward()
        self.asp.forward()
        E2 = self.pooled.shape[1]
        ops.bn_fwd(self.pooled, E2, self.bn_work, self.bn_mr, self.bn_running, st.p(FE + "asp_bn.norm.weight"),
                   st.p(FE + "asp_bn.norm.bias"), self.e2, E2, B, E2, BN_EPS, BN_MOMENTUM, False, self.train)
        ops.skinny_linear_fwd(self.e2, self.Wfc, st.p(FE + "fc.conv.bias"), self.emb, ops.ACT_NONE)
        return self.emb

    def head_forward_backward(self, label: torch.Tensor):
        return self.head.forward_backward(label)

    def backward(self) -> None:
        """Backward of embed() from the head's d(loss)/d(emb): every parameter gradient into store.grad."""
        assert self.train
        st, B = self.store, self.B
        C = self.cfg.channels
        E2, L = self.pooled.shape[1], self.cfg.lin_neurons
        ops.skinny_linear_bwd_w(self.head.demb, None, self.e2, st.g(FE + "fc.conv.weight").view(L, E2),
                                st.g(FE + "fc.conv.bias"), ops.ACT_NONE, True)
        ops.skinny_linear_bwd_x(self.head.demb, None, self.Wfc, self.de2, ops.ACT_NONE)
        ops.bn_bwd(self.de2, E2, self.pooled, E2, self.bn_mr, st.p(FE + "asp_bn.norm.weight"), self.bn_work,
                   st.g(FE + "asp_bn.norm.weight"), st.g(FE + "asp_bn.norm.bias"), self.dpooled, E2, B, E2, False)
        self.asp.backward(self.dpooled)                                          # -> d_mfa (written)
        self.mfa.backward(self.d_mfa, C[-1], self.d_cat, C[-1], False, defer_dw=True)
        nb = len(self.blocks)
        for i in range(nb, 0, -1):                                                # block i reads block i-1's output
            blk = self.blocks[i - 1]
            dout = self.d_cat[:, (i - 1) * C[1]:]
            if i == 1:
                blk.backward(dout, C[-1], self.d_x0, C[0], False)
            else:
                blk.backward(dout, C[-1], self.d_cat[:, (i - 2) * C[1]:], C[-1], True)
        self.block0.backward(self.d_x0, C[0], None, 0, False, defer_dw=True)
        # every da / conv input is final (each TDNN owns its buffers): the weight + bias gradients of ALL grouped TDNN
        # blocks in one launch (29 problems, ~550 tiles of equal length: three full rounds of the chip instead of one
        # third-filled round per SE-Res2Net block), then the tap-major -> torch layout unpack of the k > 1 kernels
        tds = [t for t in self._tdnns() if t.grouped]
        if tds:
            if self._wgs is None:
                tds.sort(key=lambda t: -t.cout * t.K)
                probs = [t.wg_problem for t in tds]
                self._wgs = [ops.WgradGroup(probs[i:i + 32], tds[0].M, tds[0].da._full.shape[0])
                             for i in range(0, len(probs), 32)]
            for wg in self._wgs:
                wg()
            for t in tds:
                t.finish_weight_grad()


class EcapaTrainer:
    """One training step: forward, AAM head, backward, fused Adam (ref: speaker_recognition_module.py:207-220)."""

    def __init__(self, store: EcapaStore, plan: EcapaPlan, schedule, process_group=None):
        self.store, self.plan, self.schedule, self.step = store, plan, schedule, 0
        self.pg = process_group                      # data parallel: ONE all-reduce of the flat gradient arena (25 MB)
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)

    def train_step(self, feat: torch.Tensor, label: torch.Tensor):
        self.store.zero_grad()
        self.plan.embed(feat)
        loss, softmax = self.plan.head_forward_backward(label)
        self.plan.backward()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.store.grad, group=self.pg)
        lr, beta1 = self.schedule.at(self.step)
        self.store.adam_step(lr, beta1, grad_scale=1.0 / self.world)
        self.step += 1
        return loss, softmax
