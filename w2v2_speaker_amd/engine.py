"""Static execution plans for the wav2vec2 speaker path on one MI355X.

A ``Plan`` is built once per (batch, samples, mode): it allocates every activation / saved-for-backward
buffer out of HBM up front (the whole base model at B=66 needs ~3 GB of 288 GB) and pre-builds every
GEMM descriptor over those fixed buffers, so a training step is a flat sequence of C-ABI launches on
one HIP stream with no allocation, no autograd graph and no host<->device sync.  Forward and the
hand-written backward follow SURVEY.md 3.2 / 8(a) rows a2-a14 (HF:382-726 +
ref: src/layers/pooling.py, src/optim/loss/aam_softmax.py, src/optim/loss/cross_entropy.py).
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import ops
from .config import W2V2Config, Wav2Vec2RegularisationConfig
from .ops import (EPI_ADD, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD, EPI_GELU_BWD, EPI_MUL, EPI_NONE, EPI_SCALE_RC, Gemm,
                  POOL_MODES,
                  WgradGroup)
from .params import W2V_PREFIX, ParamStore

_SITE = {"featproj": 1, "prologue": 2, "attn": 3, "post_attn": 4, "ffn": 5, "act": 6}


def _splitk(m_tiles: int, n_tiles: int, k: int, batch: int = 1) -> int:
    """Split K of the weight-gradient GEMMs (K = tokens) until ~2 workgroups per CU are in flight."""
    blocks = max(1, m_tiles * n_tiles * batch)
    s = max(1, min(16, 512 // blocks))
    while s > 1 and k // s < 256:
        s -= 1
    return s


def _tiles(n: int, t: int = 128) -> int:
    return (n + t - 1) // t


def encoder_backward_schedule(num_layers: int, skip: Sequence[int], pair_uppers: Sequence[int], grouped: bool = True,
                              group: int = 2):
    """Order of the encoder part of one backward pass, as plain data (no device work): a list of events
        ("body", l)        data-gradient chain of layer l up to and including the attention backward; its LayerNorm
                           gamma/beta partials are parked in the fold group; without the grouped weight-gradient
                           kernel (f32 mode) it also writes the layer's weight gradients
        ("wgrad", layers)  one grouped weight-gradient launch writing dW / dbias of `layers` (one layer, or a pair
                           (upper, upper - 1) whose dY buffers live in the two alternating scratch sets)
        ("dx", l)          G += DQKV Wqkv of layer l
        ("fold",)          gamma/beta of every LayerNorm parked since the last fold are added to the gradients
        ("notify", l)      gradient bucket "layer l" is final (LayerDrop-skipped layers: zero gradient, final at once)
    `pair_uppers` = upper layers of the weight-gradient pairs (Plan.g_wgrad_pair); `skip` = this step's LayerDrop
    decisions.  Plan.backward executes exactly this list; tests/test_host_cpu.py checks its invariants (every bucket
    is notified once, in descending order, after its last writer) for every skip pattern."""
    if grouped and group > 2:
        # Groups of `group` NON-SKIPPED layers, counted down from the top (round 6: wav2vec2-large, whose 192 tiles of
        # 256 x 256 per layer fill 256 CUs badly in ones -- 0.75 -- and in pairs -- 1.5 rounds -- but well in fours: 3.0).
        # Every layer has its OWN dY scratch set in this mode (3.5 GB of the 288 at wav2vec2-large), so a group may span
        # LayerDrop-skipped layers and its launch may come after the data-gradient products of its last member.  Buckets
        # are notified once, in descending order, after their last writer: a skipped layer inside an open group waits for
        # the group's launch.
        ev, held, skip, pending_top = [], [], set(skip), num_layers - 1

        def flush(down_to):
            nonlocal held, pending_top
            if held:
                ev.extend([("wgrad", tuple(held)), ("fold",)])
            ev.extend(("notify", h) for h in range(pending_top, down_to - 1, -1))
            held, pending_top = [], down_to - 1
        for l in reversed(range(num_layers)):
            if l in skip:
                if not held:
                    flush(l)                      # nothing open: the zero-gradient bucket is final at once
                continue
            ev.append(("body", l))
            held.append(l)
            ev.append(("dx", l))
            if len(held) == group:
                flush(l)
        flush(0)
        return ev
    ev, held, Ltop, skip, pair_uppers = [], None, num_layers - 1, set(skip), set(pair_uppers)
    for l in reversed(range(num_layers)):
        upper = grouped and (Ltop - l) % 2 == 0 and l > 0 and l in pair_uppers      # l pairs with l - 1
        if l in skip:
            if held is not None:          # partner skipped by LayerDrop: the held layer goes alone
                ev += [("wgrad", (held,)), ("fold",), ("notify", held)]
                held = None
            ev.append(("notify", l))
            continue
        ev.append(("body", l))
        if grouped:
            if held is not None:
                ev.append(("wgrad", (held, l)))
            elif not upper:
                ev.append(("wgrad", (l,)))
        ev.append(("dx", l))
        if grouped and held is None and upper:
            held = l                      # wait for layer l - 1 (its dY's live in the other buffer set)
            continue
        ev.append(("fold",))
        if held is not None:
            ev.append(("notify", held))
            held = None
        ev.append(("notify", l))
    assert held is None
    return ev


@dataclass
class LayerBufs:
    qkv: torch.Tensor
    ctx: torch.Tensor
    a: torch.Tensor        # out-proj output, overwritten with s1 = x + drop(a)
    x1: torch.Tensor
    hpre: torch.Tensor     # gelu'(FFN pre-activation), written by the FFN1 epilogue (training plans)
    h: torch.Tensor
    f: torch.Tensor        # FFN output, overwritten with s2
    mean1: torch.Tensor
    rstd1: torch.Tensor
    mean2: torch.Tensor
    rstd2: torch.Tensor
    lse: Optional[torch.Tensor] = None      # fused attention
    p: Optional[torch.Tensor] = None        # unfused attention probabilities
    pd: Optional[torch.Tensor] = None       # ... after dropout


class Plan:
    def __init__(self, store: ParamStore, batch: int, n_samples: int, *, train: bool,
                 reg: Optional[Wav2Vec2RegularisationConfig] = None, pooling: str = "mean+std",
                 insert_cls_token: bool = False, cls_token_constant: float = 1.0,
                 aam_margin: float = 0.2, aam_scale: float = 30.0, fused_attention: Optional[bool] = None,
                 seed: int = 7, keep_hidden_states: bool = False, paired: bool = False,
                 sep_token_constant: float = -1.0):
        cfg = store.cfg
        self.store, self.cfg, self.B, self.N, self.train = store, cfg, batch, n_samples, train
        # paired input (ref: wav2vec2_paired_input.py:163-207): the conv stack and projection run on 2B waveforms (first
        # B = left, last B = right utterance of each pair); the encoder sees B sequences [CLS] left [SEP] right [SEP]
        self.paired, self.sep_c = paired, sep_token_constant
        self.Bc = 2 * batch if paired else batch
        assert not (paired and insert_cls_token)
        self.reg = reg if reg is not None else Wav2Vec2RegularisationConfig()
        # "random" (IndexPool1D, ref: src/layers/pooling.py:125-126,150-154): one frame index per forward call, drawn on
        # the host with random.randint like the reference.  "none" (NoPooling, :160-166): no pooling, every frame is an
        # embedding -- the head then sees B * T rows and the label of an utterance repeated for each of its frames
        # (ref: speaker_recognition_module.py:246-267 _train_step_ce_loss_no_pooling)
        pooling = pooling.lower() if pooling.lower() == "none" else pooling
        self.pooling, self.pool_mode = pooling, POOL_MODES.get(pooling, -1)
        assert pooling in ("attentive", "random", "none") or pooling in POOL_MODES, pooling
        self.no_pool = pooling == "none"
        self._rand_idx = 0
        self.cls, self.cls_c = insert_cls_token, cls_token_constant
        self.margin, self.scale = aam_margin, aam_scale
        self.seed = seed
        # eval plans normally ping-pong two activation buffers; keeping all L+1 of them gives HF's
        # ``output_hidden_states`` (ref: wav2vec2_fc.py:440-463 ensemble of layers)
        self.all_x = train or keep_hidden_states
        self.dev, self.adt = store.device, store.act_dtype
        self.lens = cfg.conv_lengths(n_samples)
        self.T0 = self.lens[-1]                       # frames out of the CNN
        self.T = 2 * self.T0 + 3 if paired else self.T0 + (1 if insert_cls_token else 0)
        self.M0, self.M = self.Bc * self.T0, batch * self.T
        H, d = cfg.hidden_size, cfg.head_dim
        if fused_attention is None:
            fused_attention = ops.is16(self.adt) and d == 64        # any T (tiled kernels beyond 160 frames)
        self.fused = fused_attention
        # checkpoint families (config.W2V2Config): pre-LN encoder ("stable layer norm", HF:611-654,729-802) and LayerNorm
        # convolutions with bias (HF:275-299) -- the "-lv60" / xlsr models the reference can be pointed at by id
        self.stable = bool(cfg.do_stable_layer_norm)
        self.ln_conv = cfg.feat_extract_norm == "layer"
        if cfg.feat_extract_norm not in ("group", "layer"):
            raise ValueError(f"feat_extract_norm must be 'group' or 'layer', got {cfg.feat_extract_norm!r}")
        if (self.ln_conv or cfg.conv_bias) and train and not store.freeze_cnn:
            raise NotImplementedError("the layer-norm / biased convolution stack has a forward path only: train it with the "
                                      "feature extractor frozen (completely_freeze_feature_extractor: true, the reference's default)")
        if self.stable and (insert_cls_token or paired or keep_hidden_states):
            raise NotImplementedError("do_stable_layer_norm: the CLS-token, paired-input and hidden-state-ensemble paths are "
                                      "built for the post-LN encoder only")
        self.embed_dim = H * (2 if pooling == "attentive" else ops.POOL_WIDTH.get(self.pool_mode, 1))
        self._pack_version = -1
        self._cnn_version = -1
        self._asp = {}
        self.grouped = False
        self._alloc()
        self._build_gemms()

    def _wgrad_group_size(self) -> int:
        """How many consecutive blocks share one grouped weight-gradient launch: 2 (pairs, since round 3) unless FOUR fill
        the chip's rounds of 256 x 256 tiles strictly better (wav2vec2-large: 192 tiles per block -> pairs 384 = 1.5
        rounds, fours 768 = 3.0).  W2V2_WGRAD_GROUP forces 1 / 2 / 4 (tests, A/B)."""
        env = os.environ.get("W2V2_WGRAD_GROUP")
        if env:
            return max(1, min(4, int(env)))
        cfg = self.cfg
        H, I = cfg.hidden_size, cfg.intermediate_size
        c = lambda n: -(-n // 256)
        tiles = c(H) * c(I) * 2 + c(H) * c(H) + c(3 * H) * c(H)
        ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count if self.dev.type == "cuda" else 256
        cost = {k: -(-k * tiles // ncu) / k for k in (2, 4)}
        return 4 if (cfg.num_hidden_layers >= 4 and cost[4] < cost[2] - 1e-9) else 2

    def _wgrad_for(self, layers):
        """The grouped weight-gradient launch covering `layers` (a tuple from encoder_backward_schedule): prebuilt for
        single blocks and for the regular pairs, built on first use (and kept) for every other subset."""
        if len(layers) == 1:
            return self.g_layer[layers[0]]["wgrad"]
        if len(layers) == 2 and layers[0] in self.g_wgrad_pair and layers[1] == layers[0] - 1:
            return self.g_wgrad_pair[layers[0]]
        if layers not in self._wgrad_cache:
            if len(self._wgrad_cache) >= 256:          # LayerDrop patterns: bounded (descriptors only, a few KB each)
                self._wgrad_cache.clear()
            probs = []
            for l in layers:
                probs += self.g_layer[l]["wgrad_problems"]
            self._wgrad_cache[layers] = WgradGroup(probs, self.M, (self.M + 63) // 64 * 64)
        return self._wgrad_cache[layers]

    # ------------------------------------------------------------------------------------------ buffers
    def _e(self, *shape, dtype=None) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype or self.adt, device=self.dev)

    def _ep(self, rows: int, cols: int, pad_rows: int = 0) -> torch.Tensor:
        """[rows, cols] activation whose storage is padded with ZERO rows up to a multiple of 64 (at least
        ``pad_rows``): it can be the K-major operand of the grouped weight-gradient GEMM (csrc/wgrad.hip contract)."""
        rp = max((rows + 63) // 64 * 64, pad_rows)
        full = torch.zeros(rp, cols, dtype=self.adt, device=self.dev)
        v = full[:rows]
        v._w2v2_padded = full
        return v

    def _alloc(self) -> None:
        cfg, B, T, M, H, I = self.cfg, self.B, self.T, self.M, self.cfg.hidden_size, self.cfg.intermediate_size
        f32 = torch.float32
        C = cfg.conv_dim
        Bc = self.Bc
        self.stats0 = ops.conv0_workspace(Bc, self.N, C[0], cfg.conv_kernel[0], cfg.conv_stride[0], self.dev)
        self.conv = [self._e(Bc, L, c) for L, c in zip(self.lens, C)]
        cins = (1,) + tuple(C[:-1])
        self.convw = [None] + [self._e(C[i], cfg.conv_kernel[i] * cins[i]) for i in range(1, len(C))]
        self.zero_bias = torch.zeros(max(max(C), H), dtype=f32, device=self.dev)
        self.cnn_train = self.train and not self.store.freeze_cnn
        if self.cnn_train:      # unfrozen feature extractor: pre-activations saved, gradient + im2col scratch
            self.conv_pre = [None] + [self._e(Bc, L, c) for L, c in zip(self.lens[1:], C[1:])]
            self.dconv = [self._e(Bc, L, c) for L, c in zip(self.lens[:-1], C[:-1])]      # d(conv[i] output), i < last
            self.col = self._e(max(Bc * self.lens[i] * cfg.conv_kernel[i] * cins[i] for i in range(1, len(C))))
            self.dwp = torch.zeros(max(C[i] * cfg.conv_kernel[i] * cins[i] for i in range(1, len(C))), dtype=f32,
                                   device=self.dev)
            self.sums0 = self._e(Bc, C[0], 2, dtype=f32)
        # The projection's weight gradient is ONE small problem (768 x 512 outputs = 12 tiles) over all M0 tokens: alone
        # in a grouped launch it ran on 12 CUs for 144 us.  Its token dimension is cut into PROJ_SLICES slices of
        # proj_slice_len rows (a multiple of 64), each a problem of its own writing its own partial (96 workgroups),
        # summed in a fixed order afterwards -- which needs the two operands zero-padded up to slices * slice_len rows
        self.proj_slices = 8
        self.proj_slice_len = -(-self.M0 // (self.proj_slices * 64)) * 64
        proj_pad = self.proj_slices * self.proj_slice_len
        self.ln_feat = self._ep(self.M0, C[-1], proj_pad)
        self.mean_f, self.rstd_f = self._e(self.M0, dtype=f32), self._e(self.M0, dtype=f32)
        self.h0 = self._e(self.M0, H)                          # projection output (pre-CLS)
        self.hx = self._e(M, H) if (self.cls or self.paired) else self.h0       # encoder input
        G, K = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        self.Cg, self.Tp = H // G, T + K - 1
        # the direct convolution kernel is built for 48 (w2v2-base) and 64 (wav2vec2-large, round 6) channels per group at
        # 128 taps; everything else (the tiny test geometry, the exact-f32 mode) runs the implicit GEMM
        self.pos_direct = (ops.is16(self.adt) and self.Cg in (48, 64) and K == 128 and not os.environ.get("W2V2_NO_POSCONV_DIRECT"))
        self.xg = self._e(B, G, self.Tp, self.Cg)
        self.posw_f, self.posw_b = self._e(G, self.Cg, K * self.Cg), self._e(G, self.Cg, K * self.Cg)
        self.pos_sumsq = ops.weightnorm_scratch(H, G, K, self.dev)
        self.pos = self._e(M, H)                               # GELU(posconv) -> overwritten with s0
        self.pos_pre = self._e(M, H) if self.train else None
        self.mean0, self.rstd0 = self._e(M, dtype=f32), self._e(M, dtype=f32)
        L = cfg.num_hidden_layers
        nset = L if self.train else 1
        self.X = [self._ep(M, H) for _ in range(L + 1 if self.all_x else 2)]
        heads = cfg.num_attention_heads
        self.Tl = (T + 7) // 8 * 8
        self.lb: List[LayerBufs] = []
        for _ in range(nset):
            lb = LayerBufs(qkv=self._e(M, 3 * H), ctx=self._ep(M, H), a=self._e(M, H), x1=self._ep(M, H),
                           hpre=self._e(M, I) if self.train else None, h=self._ep(M, I), f=self._e(M, H),
                           mean1=self._e(M, dtype=f32), rstd1=self._e(M, dtype=f32),
                           mean2=self._e(M, dtype=f32), rstd2=self._e(M, dtype=f32))
            if self.fused:
                lb.lse = self._e(B * heads * T, dtype=f32)
            else:
                lb.p = torch.zeros(B * heads * T, self.Tl, dtype=self.adt, device=self.dev)
                lb.pd = (torch.zeros(B * heads * T, self.Tl, dtype=self.adt, device=self.dev)
                         if (self.train and self.reg.attention_dropout > 0) else lb.p)
            self.lb.append(lb)
        if not self.fused:
            self.S = torch.zeros(B * heads * T, self.Tl, dtype=f32, device=self.dev)   # scores / dP scratch
        E = self.embed_dim
        self.head_rows = B * T if self.no_pool else B          # rows the head sees
        self.emb = self._e(self.head_rows, E, dtype=f32)
        st = self.store
        self.head, self.fc = None, None
        if st.head == "bce":
            from .heads import BceHead
            self.head = BceHead(self.head_rows, E, w=st.p("linear.weight"), b=st.p("linear.bias"),
                                w_grad=st.g("linear.weight") if self.train else None,
                                b_grad=st.g("linear.bias") if self.train else None, emb=self.emb, train=self.train,
                                loss_scale=st.scaler)
        elif st.head is not None:
            from .heads import ClassifierHead, FcStack
            aam = st.head == "aam"
            nh = len(st.hidden_fc)
            # hidden Linear+ReLU layers between the pooled embedding and the head (ref: wav2vec2_fc.py:185-228)
            self.fc = FcStack(st, self.head_rows, E, st.hidden_fc, self.emb, self.train) if nh else None
            head_in = self.fc.x[-1] if self.fc is not None else self.emb
            wname = "loss_fn.fc_weights" if aam else f"fc_list.{nh}.0.weight"
            bname = f"fc_list.{nh}.0.bias"
            self.head = ClassifierHead(st.head, self.head_rows, st.head_in_dim, st.num_speakers, w_master=st.p(wname),
                                       w_operand=st.w(wname), w_grad=st.g(wname) if self.train else None,
                                       bias=None if aam else st.p(bname),
                                       bias_grad=None if (aam or not self.train) else st.g(bname),
                                       emb=head_in, act_dtype=self.adt, train=self.train, margin=self.margin,
                                       scale=self.scale, loss_scale=st.scaler)
        if self.train:
            self.demb = (self.head.demb if (self.head is not None and getattr(self, "fc", None) is None)
                         else self._e(self.head_rows, E, dtype=f32))
            self.G = self._ep(M, H, 0 if (self.cls or self.paired) else proj_pad)           # running activation gradient
            # Gradient scratch of one layer's backward: Gd = df (after the dropout mask of the FFN residual branch),
            # Gd1 = da (attention residual branch), DH, DQKV.  TWO sets, used by alternating layers: the grouped
            # weight-gradient launch then covers a PAIR of layers (8 problems = 216 tiles of 256x256, one full round
            # of the chip instead of two 216-tile rounds of 256x128), see _build_gemms / backward.
            # two alternating sets for single / paired launches; groups of four get one set PER LAYER (they may span
            # LayerDrop-skipped layers: encoder_backward_schedule)
            self.wg_group = self._wgrad_group_size()
            self._gsets = [dict(Gd=self._ep(M, H), Gd1=self._ep(M, H), DH=self._ep(M, I), DQKV=self._ep(M, 3 * H))
                           for _ in range(cfg.num_hidden_layers if self.wg_group > 2 else 2)]
            self.DC = self._e(M, H)
            self.P1 = self._e(M, H)
            self.GR = self._e(M, H) if self.stable else None        # pre-LN: gradient of the un-normalised residual stream
            self.dyg = self._e(B, G, self.Tp, self.Cg)
            self.dwf = self._e(G, K * self.Cg, self.Cg, dtype=f32)
            self.pos_dot = ops.weightnorm_scratch(H, G, K, self.dev)
            self.dn = self._e(self.M0, C[-1])
            self.G0 = self._ep(self.M0, H, proj_pad) if (self.cls or self.paired) else None
            if self.fused:
                self.delta = self._e(B * heads * T, dtype=f32)
            else:
                self.dS = torch.zeros(B * heads * T, self.Tl, dtype=self.adt, device=self.dev)

    # ------------------------------------------------------------------------------------------ descriptors
    def _build_gemms(self) -> None:
        cfg, st, B, T, M, H, I = self.cfg, self.store, self.B, self.T, self.M, self.cfg.hidden_size, self.cfg.intermediate_size
        C = cfg.conv_dim
        cins = (1,) + tuple(C[:-1])
        # conv layers 1..6: implicit GEMM over the channels-last activation of the previous layer
        self.g_conv = []
        for i in range(1, len(C)):
            k, s, ci, co = cfg.conv_kernel[i], cfg.conv_stride[i], cins[i], C[i]
            cbias = st.mp(f"feature_extractor.conv_layers.{i}.conv.bias") if cfg.conv_bias else self.zero_bias
            # layer-norm family: the product only adds the bias; LayerNorm + GELU follow in place (forward())
            self.g_conv.append(Gemm(self.Bc * self.lens[i], co, k * ci, self.conv[i - 1], self.convw[i], self.conv[i],
                                    lda=s * ci, ldb=k * ci, ldc=co, a_seg=(self.lens[i], self.lens[i - 1] * ci),
                                    epilogue=EPI_BIAS if self.ln_conv else EPI_BIAS_GELU, bias=cbias,
                                    aux=self.conv_pre[i] if self.cnn_train else None, ldaux=co))
        if self.cnn_train:
            # conv layer i backward: dW (packed layout, f32 scratch) and the im2col-space data gradient
            self.g_conv_dw, self.g_conv_dx = [None], [None]
            for i in range(1, len(C)):
                k, sd, ci, co = cfg.conv_kernel[i], cfg.conv_stride[i], cins[i], C[i]
                Mi = self.Bc * self.lens[i]
                gout = self.dn if i == len(C) - 1 else self.dconv[i]          # d(conv[i] output) -> dpre in place
                self.g_conv_dw.append(Gemm(co, k * ci, Mi, gout, self.conv[i - 1], self.dwp, lda=co, ldb=sd * ci,
                                           ldc=k * ci, transA=True, transB=True,
                                           b_seg=(self.lens[i], self.lens[i - 1] * ci),
                                           split_k=_splitk(_tiles(co), _tiles(k * ci), Mi), accumulate=True))
                self.g_conv_dx.append(Gemm(Mi, k * ci, co, gout, self.convw[i], self.col, lda=co, ldb=k * ci,
                                           ldc=k * ci, transB=True))
        mw, mp = st.mw, st.mp
        self.g_proj = Gemm(self.M0, H, C[-1], self.ln_feat, mw("feature_projection.projection.weight"), self.h0,
                           lda=C[-1], ldb=C[-1], ldc=H, epilogue=EPI_BIAS, bias=mp("feature_projection.projection.bias"))
        G, K, Cg, Tp = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings, self.Cg, self.Tp
        self.g_pos = Gemm(M, Cg, K * Cg, self.xg, self.posw_f, self.pos, lda=Cg, ldb=K * Cg, ldc=H,
                          a_seg=(T, G * Tp * Cg), batch=G, batch_inner=G, a_strides=(0, Tp * Cg),
                          b_strides=(0, Cg * K * Cg), c_strides=(0, Cg), epilogue=EPI_BIAS_GELU,
                          bias=mp("encoder.pos_conv_embed.conv.bias"), bias_stride1=Cg,
                          aux=self.pos_pre, ldaux=H, aux_strides=(0, Cg))
        heads, d = cfg.num_attention_heads, cfg.head_dim
        L = cfg.num_hidden_layers
        self.g_layer: List[Dict[str, Gemm]] = []
        for l in range(L):
            lb = self.lb[l if self.train else 0]
            xin = self.X[l] if self.all_x else self.X[l % 2]
            pre = f"encoder.layers.{l}."
            gl: Dict[str, Gemm] = {}
            # fp16: the value and output projections run with two-term weights (W = hi + lo planes, 2 x K steps on
            # those columns): their rounding is the largest single term of the embedding error (DESIGN "precision")
            two = st.two_term and H % 64 == 0
            gl["qkv"] = Gemm(M, 3 * H, H, xin, st.qkv(l, "w"), lb.qkv, lda=H, ldb=H, ldc=3 * H, epilogue=EPI_BIAS,
                             bias=st.qkv(l, "p", "bias"), b_lo=st.qkv_lo(l) if two else None, n_ext_from=2 * H)
            if not self.fused:
                qkv = lb.qkv.view(-1)
                sc = (heads * T * self.Tl, T * self.Tl)
                gl["scores"] = Gemm(T, T, d, qkv, qkv[H:], self.S, lda=3 * H, ldb=3 * H, ldc=self.Tl,
                                    batch=B * heads, batch_inner=heads, a_strides=(T * 3 * H, d),
                                    b_strides=(T * 3 * H, d), c_strides=sc, alpha=d ** -0.5)
                gl["ctx"] = Gemm(T, d, T, lb.pd, qkv[2 * H:], lb.ctx, lda=self.Tl, ldb=3 * H, ldc=H, transB=True,
                                 batch=B * heads, batch_inner=heads, a_strides=sc, b_strides=(T * 3 * H, d),
                                 c_strides=(T * H, d))
            gl["out"] = Gemm(M, H, H, lb.ctx, mw(pre + "attention.out_proj.weight"), lb.a, lda=H, ldb=H, ldc=H,
                             epilogue=EPI_BIAS, bias=mp(pre + "attention.out_proj.bias"),
                             b_lo=st.w_lo(W2V_PREFIX + pre + "attention.out_proj.weight") if two else None)
            # training: the epilogue leaves gelu'(pre) in lb.hpre (the f32 pre-activation is in registers there), the
            # backward product "dh" just multiplies by it (w2v2_hip.h, W2V2_EPI_BIAS_GELU_GRAD / W2V2_EPI_MUL)
            gl["ffn1"] = Gemm(M, I, H, lb.x1, mw(pre + "feed_forward.intermediate_dense.weight"), lb.h, lda=H, ldb=H,
                              ldc=I, epilogue=EPI_BIAS_GELU_GRAD if lb.hpre is not None else EPI_BIAS_GELU,
                              bias=mp(pre + "feed_forward.intermediate_dense.bias"), aux=lb.hpre, ldaux=I)
            gl["ffn2"] = Gemm(M, H, I, lb.h, mw(pre + "feed_forward.output_dense.weight"), lb.f, lda=I, ldb=I, ldc=H,
                              epilogue=EPI_BIAS, bias=mp(pre + "feed_forward.output_dense.bias"))
            if self.train:
                mg = st.mg
                sk = lambda m, n: _splitk(_tiles(m), _tiles(n), M)
                W2, W1, Wo = (mw(pre + "feed_forward.output_dense.weight"), mw(pre + "feed_forward.intermediate_dense.weight"),
                              mw(pre + "attention.out_proj.weight"))
                Wqkv = st.qkv(l, "w")
                # data-gradient products: with the pre-transposed bf16 weight copies they are plain
                # K-contiguous GEMMs (LDS-DMA kernel); the f32 parity mode reads W as a K-major operand
                gs = self._gsets[l % len(self._gsets)]
                tb = st.flat_lp_t is None
                if not tb:
                    W2, W1, Wo = (st.wt("wav2vec.model." + pre + "feed_forward.output_dense.weight"),
                                  st.wt("wav2vec.model." + pre + "feed_forward.intermediate_dense.weight"),
                                  st.wt("wav2vec.model." + pre + "attention.out_proj.weight"))
                    Wqkv = st.qkv_t(l)
                self.grouped = st.flat_lp_t is not None      # bf16 mode: one grouped, atomic-free wgrad launch
                if self.grouped:
                    pad = lambda t: t._w2v2_padded
                    Mp = (M + 63) // 64 * 64
                    gl["wgrad_problems"] = [
                        (pad(gs["Gd"]), pad(lb.h), mg(pre + "feed_forward.output_dense.weight"),
                         mg(pre + "feed_forward.output_dense.bias")),
                        (pad(gs["DH"]), pad(lb.x1), mg(pre + "feed_forward.intermediate_dense.weight"),
                         mg(pre + "feed_forward.intermediate_dense.bias")),
                        (pad(gs["Gd1"]), pad(lb.ctx), mg(pre + "attention.out_proj.weight"),
                         mg(pre + "attention.out_proj.bias")),
                        (pad(gs["DQKV"]), pad(xin), st.qkv(l, "g"), st.qkv(l, "g", "bias"))]
                    gl["wgrad"] = WgradGroup(gl["wgrad_problems"], M, Mp)
                gl["dW2"] = Gemm(H, I, M, gs["Gd"], lb.h, mg(pre + "feed_forward.output_dense.weight"), lda=H, ldb=I,
                                 ldc=I, transA=True, transB=True, split_k=sk(H, I), accumulate=True)
                gl["dh"] = Gemm(M, I, H, gs["Gd"], W2, gs["DH"], lda=H, ldb=I if tb else H, ldc=I, transB=tb,
                                epilogue=EPI_MUL, aux=lb.hpre, ldaux=I)
                gl["dW1"] = Gemm(I, H, M, gs["DH"], lb.x1, mg(pre + "feed_forward.intermediate_dense.weight"), lda=I,
                                 ldb=H, ldc=H, transA=True, transB=True, split_k=sk(I, H), accumulate=True)
                # post-LN: G = DH @ W1 + G (the residual path rides in G); pre-LN: G = DH @ W1 alone (the residual
                # gradient lives in self.GR, _backward_encoder_stable)
                gl["dx1"] = Gemm(M, H, I, gs["DH"], W1, self.G, lda=I, ldb=H if tb else I, ldc=H, transB=tb,
                                 epilogue=EPI_NONE if self.stable else EPI_ADD, aux=None if self.stable else self.G,
                                 ldaux=0 if self.stable else H)
                gl["dWo"] = Gemm(H, H, M, gs["Gd1"], lb.ctx, mg(pre + "attention.out_proj.weight"), lda=H, ldb=H, ldc=H,
                                 transA=True, transB=True, split_k=sk(H, H), accumulate=True)
                gl["dctx"] = Gemm(M, H, H, gs["Gd1"], Wo, self.DC, lda=H, ldb=H, ldc=H, transB=tb)
                if not self.fused:
                    qkv = lb.qkv.view(-1)
                    dq = gs["DQKV"].view(-1)
                    sc = (heads * T * self.Tl, T * self.Tl)
                    hs = (T * 3 * H, d)
                    gl["dP"] = Gemm(T, T, d, self.DC, qkv[2 * H:], self.S, lda=H, ldb=3 * H, ldc=self.Tl,
                                    batch=B * heads, batch_inner=heads, a_strides=(T * H, d), b_strides=hs, c_strides=sc)
                    gl["dq"] = Gemm(T, d, T, self.dS, qkv[H:], dq, lda=self.Tl, ldb=3 * H, ldc=3 * H, transB=True,
                                    batch=B * heads, batch_inner=heads, a_strides=sc, b_strides=hs, c_strides=hs,
                                    alpha=d ** -0.5)
                    gl["dk"] = Gemm(T, d, T, self.dS, qkv, dq[H:], lda=self.Tl, ldb=3 * H, ldc=3 * H, transA=True,
                                    transB=True, batch=B * heads, batch_inner=heads, a_strides=sc, b_strides=hs,
                                    c_strides=hs, alpha=d ** -0.5)
                    gl["dv"] = Gemm(T, d, T, lb.pd, self.DC, dq[2 * H:], lda=self.Tl, ldb=H, ldc=3 * H, transA=True,
                                    transB=True, batch=B * heads, batch_inner=heads, a_strides=sc,
                                    b_strides=(T * H, d), c_strides=hs)
                gl["dWqkv"] = Gemm(3 * H, H, M, gs["DQKV"], xin, st.qkv(l, "g"), lda=3 * H, ldb=H, ldc=H, transA=True,
                                   transB=True, split_k=sk(3 * H, H), accumulate=True)
                gl["dx"] = Gemm(M, H, 3 * H, gs["DQKV"], Wqkv, self.G, lda=3 * H, ldb=H if tb else 3 * H, ldc=H,
                                transB=tb, epilogue=EPI_NONE if self.stable else EPI_ADD,
                                aux=None if self.stable else self.G, ldaux=0 if self.stable else H)
            self.g_layer.append(gl)
        # weight gradients of two consecutive layers (l, l-1; l counted down from the top) in one launch
        self.g_wgrad_pair = {}
        self._wgrad_cache = {}
        if self.train and getattr(self, "grouped", False) and not os.environ.get("W2V2_NO_WGRAD_PAIRS"):
            Mp = (M + 63) // 64 * 64
            for l in range(L - 1, 0, -2):
                self.g_wgrad_pair[l] = WgradGroup(self.g_layer[l]["wgrad_problems"] +
                                                  self.g_layer[l - 1]["wgrad_problems"], M, Mp)
        if self.train:
            mg = st.mg
            # pos-conv backward: weight gradient (packed layout) and data gradient (flipped weights)
            # weight gradient with the LARGE dimension (tap, ci) = 6144 as M and the 48 output channels of the
            # group as N (narrow 128x64 tiles, 75 % full) instead of M = 48 on 128-row tiles (37 % full)
            self.g_pos_dw = Gemm(K * Cg, Cg, M, self.xg, self.P1, self.dwf, lda=Cg, ldb=H, ldc=Cg, transA=True,
                                 transB=True, a_seg=(T, G * Tp * Cg), batch=G, batch_inner=G, a_strides=(0, Tp * Cg),
                                 b_strides=(0, Cg), c_strides=(0, K * Cg * Cg))
            self.g_pos_dx = Gemm(M, Cg, K * Cg, self.dyg, self.posw_b, self.G, lda=Cg, ldb=K * Cg, ldc=H,
                                 a_seg=(T, G * Tp * Cg), batch=G, batch_inner=G, a_strides=(0, Tp * Cg),
                                 b_strides=(0, Cg * K * Cg), c_strides=(0, Cg), epilogue=EPI_ADD, aux=self.G, ldaux=H,
                                 aux_strides=(0, Cg))
            g0 = self.G0 if (self.cls or self.paired) else self.G
            if st.flat_lp_t is not None:
                S, Ls, Cf = self.proj_slices, self.proj_slice_len, C[-1]
                self.proj_parts = torch.zeros(S, H * Cf + H, dtype=torch.float32, device=self.dev)   # [slice][dW | db]
                g0p, lnp = g0._w2v2_padded, self.ln_feat._w2v2_padded
                assert g0p.shape[0] >= S * Ls and lnp.shape[0] >= S * Ls
                self.g_proj_dw = WgradGroup([(g0p[i * Ls:], lnp[i * Ls:], self.proj_parts[i, :H * Cf].view(H, Cf),
                                              self.proj_parts[i, H * Cf:]) for i in range(S)], Ls, Ls)
            else:
                self.g_proj_dw = Gemm(H, C[-1], self.M0, g0, self.ln_feat, mg("feature_projection.projection.weight"),
                                      lda=H, ldb=C[-1], ldc=C[-1], transA=True, transB=True,
                                      split_k=_splitk(_tiles(H), _tiles(C[-1]), self.M0), accumulate=True)
            tbp = st.flat_lp_t is None
            Wp = (mw("feature_projection.projection.weight") if tbp
                  else st.wt("wav2vec.model.feature_projection.projection.weight"))
            self.g_proj_dn = Gemm(self.M0, C[-1], H, g0, Wp, self.dn, lda=H, ldb=C[-1] if tbp else H, ldc=C[-1],
                                  transB=tbp)

    # ------------------------------------------------------------------------------------------ derived weights
    def _refresh_packs(self) -> None:
        st, cfg = self.store, self.cfg
        if self._cnn_version != st.cnn_version:
            for i in range(1, len(cfg.conv_dim)):
                ops.pack_conv_weight(st.mp(f"feature_extractor.conv_layers.{i}.conv.weight"), self.convw[i])
            self._cnn_version = st.cnn_version
        if self._pack_version != st.version:
            ops.weightnorm_pack(st.mp("encoder.pos_conv_embed.conv.parametrizations.weight.original0"),
                                st.mp("encoder.pos_conv_embed.conv.parametrizations.weight.original1"),
                                self.pos_sumsq, self.posw_f, self.posw_b, cfg.hidden_size,
                                cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings)
            self._pack_version = st.version

    def _sd(self, site: str, layer: int, step: int) -> int:
        """Dropout stream key for (site, layer, step)."""
        return (self.seed * 1000003 + step) * 4096 + layer * 8 + _SITE[site]

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, wav: torch.Tensor, mask: Optional[torch.Tensor] = None, skip_layers: Sequence[int] = (),
                step: int = 0, feature_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """wav [B,N] (or [B,1,N]) f32 on the GPU -> last_hidden_state [B,T,H] (act dtype).
        mask: [B,T0] uint8/bool SpecAugment time mask, feature_mask: [B,H] uint8/bool SpecAugment feature mask
        (HF:1294-1304; training only).  Dropout is active iff the plan was built with train=True and the
        regularisation probabilities are > 0."""
        cfg, st, reg = self.cfg, self.store, self.reg
        B, T, M, H = self.B, self.T, self.M, cfg.hidden_size
        if wav.dim() == 3:
            wav = wav[:, 0, :]
        wav = wav.contiguous()
        assert wav.shape == (self.Bc, self.N) and wav.dtype == torch.float32 and wav.is_cuda
        self._refresh_packs()
        self._wav = wav
        tr = self.train
        self._step, self._skip, self._mask, self._fmask = step, tuple(skip_layers), None, None
        mp = st.mp
        self._cnn_forward(wav)
        ops.layernorm_fwd(self.conv[-1].view(self.M0, -1), None, mp("feature_projection.layer_norm.weight"),
                          mp("feature_projection.layer_norm.bias"), self.ln_feat, self.mean_f, self.rstd_f,
                          cfg.layer_norm_eps)
        self.g_proj()
        if tr and reg.feat_proj_dropout > 0:
            ops.dropout_(self.h0, reg.feat_proj_dropout, self._sd("featproj", 0, step))
        if self.cls:
            ops.prepend_token(self.h0.view(B, self.T0, H), self.hx.view(B, T, H), self.cls_c)
        elif self.paired:
            # [CLS] left [SEP] right [SEP] (constant tokens; no SpecAugment on this path) -- buffer plumbing only
            assert mask is None
            T0, hx, h0 = self.T0, self.hx.view(B, T, H), self.h0.view(2, B, self.T0, H)
            hx[:, 0].fill_(self.cls_c)
            hx[:, 1:1 + T0].copy_(h0[0])
            hx[:, 1 + T0].fill_(self.sep_c)
            hx[:, 2 + T0:2 + 2 * T0].copy_(h0[1])
            hx[:, 2 + 2 * T0].fill_(self.sep_c)
        elif mask is not None:
            self._mask = mask.to(torch.uint8).contiguous().view(-1)
            ops.mask_fill(self.h0, self._mask, mp("masked_spec_embed"))
        if feature_mask is not None:          # after the time mask, like HF's _mask_hidden_states
            assert not (self.cls or self.paired)
            self._fmask = feature_mask.to(torch.uint8).contiguous().view(-1)
            ops.mask_feature(self.h0, self._fmask, B, self.T0)
        G, K = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        ops.posconv_regroup(self.hx, self.xg, B, T, H, G, K, K // 2)
        if self.pos_direct:      # image-resident direct convolution (csrc/posconv_direct.hip), bit-equal to the GEMM
            ops.posconv_direct(self.xg, self.posw_f, self.pos, self.pos_pre, mp("encoder.pos_conv_embed.conv.bias"), B, T, G,
                               self.Cg, K, H, 0)
        else:
            self.g_pos()
        if self.stable:
            return self._forward_encoder_stable(step)
        x = self.X[0]
        ops.layernorm_fwd(self.hx, self.pos, mp("encoder.layer_norm.weight"), mp("encoder.layer_norm.bias"), x,
                          self.mean0, self.rstd0, cfg.layer_norm_eps)
        if tr and reg.hidden_dropout > 0:
            ops.dropout_(x, reg.hidden_dropout, self._sd("prologue", 0, step))
        heads, d = cfg.num_attention_heads, cfg.head_dim
        pa = reg.attention_dropout if tr else 0.0
        ph = reg.hidden_dropout if tr else 0.0
        for l in range(cfg.num_hidden_layers):
            xin = self.X[l] if self.all_x else self.X[l % 2]
            xout = self.X[l + 1] if self.all_x else self.X[(l + 1) % 2]
            if l in self._skip:                 # LayerDrop (HF:698-709): the layer is the identity
                ops.copy_strided(xin, H, xout, H, xin.shape[0], H)
                continue
            lb, gl = self.lb[l if tr else 0], self.g_layer[l]
            pre = f"encoder.layers.{l}."
            gl["qkv"]()
            if self.fused:
                ops.attention_fwd(lb.qkv, lb.ctx, lb.lse, B, T, heads, d, d ** -0.5, pa, self._sd("attn", l, step))
            else:
                gl["scores"]()
                ops.softmax_fwd(self.S, lb.p, lb.pd if pa > 0 else None, B * heads * T, T, self.Tl, pa,
                                self._sd("attn", l, step))
                gl["ctx"]()
            gl["out"]()
            ops.layernorm_fwd(xin, lb.a, mp(pre + "layer_norm.weight"), mp(pre + "layer_norm.bias"), lb.x1, lb.mean1,
                              lb.rstd1, cfg.layer_norm_eps, ph, self._sd("post_attn", l, step))
            gl["ffn1"]()
            if tr and reg.activation_dropout > 0:
                # HF:566-569 intermediate_dropout.  The SAME keep decisions (same stream key, same element index) are
                # applied to the saved gelu'(pre): the backward product "dh" multiplies by it, which is then exactly
                # d(dropout(gelu(pre))) / d(pre) -- no separate dropout backward, no mask tensor
                ops.dropout_(lb.h, reg.activation_dropout, self._sd("act", l, step))
                ops.dropout_(lb.hpre, reg.activation_dropout, self._sd("act", l, step))
            gl["ffn2"]()
            ops.layernorm_fwd(lb.x1, lb.f, mp(pre + "final_layer_norm.weight"), mp(pre + "final_layer_norm.bias"),
                              xout, lb.mean2, lb.rstd2, cfg.layer_norm_eps, ph, self._sd("ffn", l, step))
        self.out = (self.X[cfg.num_hidden_layers] if self.all_x else self.X[cfg.num_hidden_layers % 2]).view(B, T, H)
        return self.out

    def _ln_params_for_input_of(self, l: int):
        """Pre-LN encoder: the LayerNorm that produces X[l], the normalised input of block l (its ``layer_norm``), or -- for
        l = number of blocks -- the encoder's final ``layer_norm`` (HF:791)."""
        n = f"encoder.layers.{l}.layer_norm" if l < self.cfg.num_hidden_layers else "encoder.layer_norm"
        return n + ".weight", n + ".bias"

    def _forward_encoder_stable(self, step: int) -> torch.Tensor:
        """Pre-LN encoder (do_stable_layer_norm, HF:611-654,729-802) on the SAME buffers, GEMM descriptors and fused
        residual + dropout + LayerNorm kernel as the post-LN one, re-wired:
            x_0 = drop(hx + pos)                                   X[0] = LN1_0(x_0)
            block l:  a = out_proj(attn(qkv(X[l])))                s1 = x_l + drop(a)     -> lb.a,   lb.x1 = LN2_l(s1)
                      f = ffn2(gelu(ffn1(lb.x1)))                  x_{l+1} = s1 + drop(f) -> lb.f,   X[l+1] = LN1_{l+1}(x_{l+1})
            (the last X is the encoder's final LayerNorm).  The un-normalised residual stream x_l lives in the tensor the
            fused kernel writes its pre-norm sum into; a LayerDrop-skipped block passes it on and only re-normalises it for
            the next block (self._res_src[l] remembers which tensor that was, for the backward)."""
        cfg, st, reg, tr = self.cfg, self.store, self.reg, self.train
        B, T, H = self.B, self.T, cfg.hidden_size
        mp, eps = st.mp, cfg.layer_norm_eps
        heads, d = cfg.num_attention_heads, cfg.head_dim
        pa = reg.attention_dropout if tr else 0.0
        ph = reg.hidden_dropout if tr else 0.0
        ops.add(self.hx, self.pos, self.pos)                      # x_0 (before dropout) replaces the positional embedding
        if ph > 0:
            ops.dropout_(self.pos, ph, self._sd("prologue", 0, step))
        res = self.pos
        w0, b0 = self._ln_params_for_input_of(0)
        ops.layernorm_fwd(res, None, mp(w0), mp(b0), self.X[0], self.mean0, self.rstd0, eps)
        self._res_src = {}
        L = cfg.num_hidden_layers
        for l in range(L):
            xin = self.X[l] if self.all_x else self.X[l % 2]
            xout = self.X[l + 1] if self.all_x else self.X[(l + 1) % 2]
            lb, gl = self.lb[l if tr else 0], self.g_layer[l]
            wn, bn = self._ln_params_for_input_of(l + 1)
            pre = f"encoder.layers.{l}."
            if l in self._skip:                 # LayerDrop: x_{l+1} = x_l, re-normalised for the next block
                ops.layernorm_fwd(res, None, mp(wn), mp(bn), xout, lb.mean2, lb.rstd2, eps)
                self._res_src[l] = res
                continue
            gl["qkv"]()
            if self.fused:
                ops.attention_fwd(lb.qkv, lb.ctx, lb.lse, B, T, heads, d, d ** -0.5, pa, self._sd("attn", l, step))
            else:
                gl["scores"]()
                ops.softmax_fwd(self.S, lb.p, lb.pd if pa > 0 else None, B * heads * T, T, self.Tl, pa,
                                self._sd("attn", l, step))
                gl["ctx"]()
            gl["out"]()
            ops.layernorm_fwd(res, lb.a, mp(pre + "final_layer_norm.weight"), mp(pre + "final_layer_norm.bias"), lb.x1,
                              lb.mean1, lb.rstd1, eps, ph, self._sd("post_attn", l, step))          # lb.a <- s1
            gl["ffn1"]()
            if tr and reg.activation_dropout > 0:
                ops.dropout_(lb.h, reg.activation_dropout, self._sd("act", l, step))
                ops.dropout_(lb.hpre, reg.activation_dropout, self._sd("act", l, step))
            gl["ffn2"]()
            ops.layernorm_fwd(lb.a, lb.f, mp(wn), mp(bn), xout, lb.mean2, lb.rstd2, eps, ph,
                              self._sd("ffn", l, step))                                              # lb.f <- x_{l+1}
            res = lb.f
        self.out = (self.X[L] if self.all_x else self.X[L % 2]).view(B, T, H)
        return self.out

    def conv_features(self, wav: torch.Tensor) -> torch.Tensor:
        """The conv feature extractor alone (HF:382-419; ref: src/models/wav2vec2.py:163-169,
        ``Wav2vecLiteWrapperModule.forward``): wav [B,N] f32 -> [B, L, 512] (act dtype, channels-last; the reference
        returns the transpose).  Same kernels as the first stage of forward()."""
        cfg, st = self.cfg, self.store
        if wav.dim() == 3:
            wav = wav[:, 0, :]
        wav = wav.contiguous()
        assert wav.shape == (self.Bc, self.N) and wav.dtype == torch.float32 and wav.is_cuda
        self._refresh_packs()
        self._wav = wav
        self._cnn_forward(wav)
        return self.conv[-1]

    def _cnn_forward(self, wav: torch.Tensor) -> None:
        """The seven convolution layers (HF:382-419) into self.conv[i] ([Bc, L_i, C_i], channels-last).  Group-norm family
        (wav2vec2-base / -large): layer 0 = conv + GroupNorm + GELU in one kernel pair, layers 1-6 = implicit GEMM with
        a bias + GELU epilogue (HF:254-272,302-323; a layer-0 conv bias would be cancelled by the GroupNorm's mean over time
        and is not read).  Layer-norm family ("-lv60" / xlsr, HF:275-299): every layer is conv (+ bias) -> LayerNorm over
        the channels -> GELU; layer 0 in one kernel, layers 1-6 as GEMM (bias epilogue) + an in-place LayerNorm-GELU pass."""
        cfg, mp = self.cfg, self.store.mp
        fe = "feature_extractor.conv_layers."
        if not self.ln_conv:
            ops.conv0_groupnorm_gelu(wav, mp(fe + "0.conv.weight"), mp(fe + "0.layer_norm.weight"), mp(fe + "0.layer_norm.bias"),
                                     self.conv[0], self.stats0, cfg.conv_kernel[0], cfg.conv_stride[0])
            for g in self.g_conv:
                g()
            return
        ops.conv0_layernorm_gelu(wav, mp(fe + "0.conv.weight"), mp(fe + "0.conv.bias") if cfg.conv_bias else None,
                                 mp(fe + "0.layer_norm.weight"), mp(fe + "0.layer_norm.bias"), self.conv[0],
                                 cfg.conv_kernel[0], cfg.conv_stride[0])
        for i, g in enumerate(self.g_conv, start=1):
            g()
            y = self.conv[i].view(-1, cfg.conv_dim[i])
            ops.layernorm_gelu_fwd(y, mp(fe + f"{i}.layer_norm.weight"), mp(fe + f"{i}.layer_norm.bias"), y, 1e-5)

    def conv_backward(self, dfeat: torch.Tensor) -> None:
        """Backward of conv_features(): dfeat [B, L, 512] = d(loss)/d(features); the gradients of the seven conv layers
        and the layer-0 GroupNorm are ACCUMULATED in the arena.  Needs a plan built with train=True over a store with
        ``freeze_cnn=False``."""
        if not self.train or self.store.freeze_cnn:
            raise RuntimeError("conv_backward needs Plan(train=True) over ParamStore(freeze_cnn=False)")
        self.dn.view(-1).copy_(dfeat.to(self.adt).contiguous().view(-1))
        self._backward_cnn()

    def embed(self, wav, mask=None, skip_layers=(), step: int = 0, feature_mask=None) -> torch.Tensor:
        """ref: src/lightning_modules/speaker/wav2vec2_fc.py:414-431 -> pooled embedding [B,E] f32."""
        out = self.forward(wav, mask, skip_layers, step, feature_mask)
        if self.pooling == "attentive":
            return self._asp_for(out).forward()
        self._pool_fwd(out)
        return self.emb

    def _pool_fwd(self, out: torch.Tensor) -> None:
        B, T, H = out.shape
        if self.no_pool:
            ops.pool_fwd(out.view(B * T, 1, H), self.emb, POOL_MODES["first"])      # f32 copy of every frame
        elif self.pooling == "random":
            import random
            self._rand_idx = random.randint(0, T - 1)                             # ref: pooling.py:150-154
            ops.pool_fwd(out, self.emb, ops.POOL_INDEX_BASE + self._rand_idx)
        else:
            ops.pool_fwd(out, self.emb, self.pool_mode)

    def hidden_states(self) -> List[torch.Tensor]:
        """HF ``output_hidden_states`` of the last forward: [prologue output, layer 1, ..., layer L], each [B,T,H]."""
        assert self.all_x, "build the plan with keep_hidden_states=True"
        return [x.view(self.B, self.T, self.cfg.hidden_size) for x in self.X]

    def ensemble_embeddings(self, wav, num_ensembles: int) -> List[torch.Tensor]:
        """ref: src/lightning_modules/speaker/wav2vec2_fc.py:440-463 (compute_ensemble_embedding): the pooled
        embedding of each of the last ``num_ensembles`` hidden states (hidden_states[L+1-n : L+1])."""
        assert self.pooling != "attentive", "ensemble pooling shares one stat-pooling layer: use a stateless pool"
        self.forward(wav, None, (), 0)
        hs = self.hidden_states()
        out = []
        for h in hs[len(hs) - num_ensembles:]:
            e = torch.empty_like(self.emb)
            ops.pool_fwd(h, e, self.pool_mode)
            out.append(e)
        return out

    def _asp_for(self, out: torch.Tensor):
        """Attentive statistics pooling over the buffer the encoder just wrote (built once per buffer)."""
        from .asp import AttentivePool
        key = out.data_ptr()
        if key not in self._asp:
            B, T, H = out.shape
            x2 = out.view(B * T, H)
            src = next((x for x in self.X if x.data_ptr() == key), None)
            if src is not None and hasattr(src, "_w2v2_padded"):
                x2._w2v2_padded = src._w2v2_padded
            self._asp[key] = AttentivePool(self.store, x2, self.emb, self.G if self.train else None, B, T, self.train)
        self._asp_cur = self._asp[key]
        return self._asp_cur

    # ------------------------------------------------------------------------------------------ head
    def head_forward_backward(self, label: torch.Tensor):
        """Loss head on self.emb (hidden FC layers, then heads.ClassifierHead): returns (loss scalar tensor,
        softmax [B,C]); a training plan also leaves d(loss)/d(pooled embedding) in self.demb and the head gradients
        in the flat buffer."""
        if self.no_pool:                 # one prediction per frame: the utterance label for each of its T frames
            label = torch.repeat_interleave(label, self.T)
        if self.fc is not None:
            self.fc.forward()
        out = self.head.forward_backward(label)
        if self.fc is not None and self.train:
            self.demb.copy_(self.fc.backward(self.head.demb))
        return out

    def speaker_embedding(self, embedding_layer_idx: int = -1) -> torch.Tensor:
        """ref: wav2vec2_fc.py:363-398 (_fc_head_ops_pre_spk_embedding) after embed(): the pooled embedding
        (idx < 0), the output of hidden layer idx, or (idx == number of hidden layers, CE head) the logits."""
        if embedding_layer_idx < 0:
            return self.emb
        nh = len(self.store.hidden_fc)
        if embedding_layer_idx < nh:
            return self.fc.forward(upto=embedding_layer_idx)
        if embedding_layer_idx == nh and self.store.head == "ce":
            if self.fc is not None:
                self.fc.forward()
            h = self.head
            if h.emb_lp is not h.emb:
                ops.cast(h.emb, h.emb_lp)
            h.g_fwd()
            return h.logits[:, :h.C]
        raise ValueError("could not determine the speaker embedding layer")

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, demb: Optional[torch.Tensor] = None,
                 on_bucket_ready: Optional[Callable[[str], None]] = None,
                 dhidden: Optional[torch.Tensor] = None) -> None:
        """Backward of embed(): demb [B,E] f32 (default self.demb from the head) -> parameter gradients
        accumulated into store.grad.  on_bucket_ready(name) fires as soon as a gradient bucket
        ('head', 'layer11', ..., 'layer0', 'prologue', 'projection') is final, for the overlapped all-reduce."""
        assert self.train, "backward needs a training plan"
        cfg, st, reg = self.cfg, self.store, self.reg
        B, T, M, H = self.B, self.T, self.M, cfg.hidden_size
        mp, mg = st.mp, st.mg
        step = self._step
        notify = on_bucket_ready or (lambda name: None)
        if dhidden is not None:           # gradient wrt last_hidden_state given directly (module surface)
            self.G.view(B, T, H).copy_(dhidden)
        else:
            if demb is None:
                demb = self.demb
            if self.pooling == "attentive":   # its parameters share the head's gradient bucket
                self._asp_cur.backward(demb)
            elif self.no_pool:
                ops.pool_bwd(self.out.view(B * T, 1, H), self.emb, demb, self.G.view(B * T, 1, H), POOL_MODES["first"])
            elif self.pooling == "random":
                ops.pool_bwd(self.out, self.emb, demb, self.G.view(B, T, H), ops.POOL_INDEX_BASE + self._rand_idx)
            else:
                ops.pool_bwd(self.out, self.emb, demb, self.G.view(B, T, H), self.pool_mode)
        notify("head")
        if getattr(self, "_lnfold", None) is None:
            self._lnfold = ops.LnFoldGroup(H, self.dev)
        lnfold = self._lnfold if not os.environ.get("W2V2_NO_LN_FOLD") else None
        if self.stable:
            self._backward_encoder_stable(notify)               # leaves d(loss)/d(hx + pos) in self.G
        for ev in (() if self.stable else
                   encoder_backward_schedule(cfg.num_hidden_layers, self._skip, self.g_wgrad_pair, self.grouped,
                                             1 if os.environ.get("W2V2_NO_WGRAD_PAIRS") else getattr(self, "wg_group", 2))):
            kind = ev[0]
            if kind == "body":
                self._layer_backward_body(ev[1], lnfold)
            elif kind == "wgrad":
                layers = ev[1]
                self._wgrad_for(tuple(layers))()
            elif kind == "dx":
                self.g_layer[ev[1]]["dx"]()                     # G = DQKV @ Wqkv + G
            elif kind == "fold":
                if lnfold is not None:
                    lnfold.fold()                               # gamma / beta of the bucket's LayerNorms: one launch
            else:
                notify(f"layer{ev[1]}")
        # encoder prologue: x0 = drop(LN(hx + pos)), pos = GELU(posconv(hx) + b)
        ph = reg.hidden_dropout
        if not self.stable:
            if ph > 0:
                ops.dropout_(self.G, ph, self._sd("prologue", 0, step))
            ops.layernorm_bwd(self.G, self.pos, self.mean0, self.rstd0, mp("encoder.layer_norm.weight"), self.G, None,
                              mg("encoder.layer_norm.weight"), mg("encoder.layer_norm.bias"))
        if H % 8 == 0:
            ops.gelu_bwd_colsum(self.G, self.pos_pre, self.P1, mg("encoder.pos_conv_embed.conv.bias"), M, H)
        else:
            ops.gelu_bwd(self.G, self.pos_pre, self.P1)
            ops.colsum(self.P1, mg("encoder.pos_conv_embed.conv.bias"), M, H)
        G_, K = cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
        if ops.is16(self.P1.dtype) and not os.environ.get("W2V2_POS_DW_GEMM"):
            ops.posconv_wgrad(self.P1, self.xg, self.dwf, B, T, H, G_, K)      # correlation kernel (posconv_wgrad.hip)
        else:
            self.g_pos_dw()                                                    # exact-f32 mode: implicit GEMM
        ops.weightnorm_bwd(mp("encoder.pos_conv_embed.conv.parametrizations.weight.original0"),
                           mp("encoder.pos_conv_embed.conv.parametrizations.weight.original1"), self.pos_sumsq,
                           self.dwf, self.pos_dot, mg("encoder.pos_conv_embed.conv.parametrizations.weight.original0"),
                           mg("encoder.pos_conv_embed.conv.parametrizations.weight.original1"), H, G_, K)
        notify("prologue")            # encoder LayerNorm, pos-conv bias and weight-norm pair are final (params.grad_buckets)
        ops.posconv_regroup(self.P1, self.dyg, B, T, H, G_, K, K - 1 - K // 2)
        if self.pos_direct:
            ops.posconv_direct(self.dyg, self.posw_b, self.G, self.G, None, B, T, G_, self.Cg, K, H, 1)
        else:
            self.g_pos_dx()                                     # G = conv^T(P1) + G
        if self.cls:
            self.G0.view(B, self.T0, H).copy_(self.G.view(B, T, H)[:, 1:, :])    # the CLS row has no input
            g0 = self.G0
        elif self.paired:                     # the three constant tokens have no input
            T0, G3, G0 = self.T0, self.G.view(B, T, H), self.G0.view(2, B, self.T0, H)
            G0[0].copy_(G3[:, 1:1 + T0])
            G0[1].copy_(G3[:, 2 + T0:2 + 2 * T0])
            g0 = self.G0
        else:
            g0 = self.G
            if self._fmask is not None:
                ops.mask_feature(g0, self._fmask, B, self.T0)
            if self._mask is not None:
                ops.mask_fill_bwd(g0, self._mask, mg("masked_spec_embed"))
        if reg.feat_proj_dropout > 0:
            ops.dropout_(g0, reg.feat_proj_dropout, self._sd("featproj", 0, step))
        self.g_proj_dw()
        if st.flat_lp_t is None:
            ops.colsum(g0, mg("feature_projection.projection.bias"), self.M0, H)
        else:          # fixed-order sum of the slices' partials (one writer per column: no atomics race, reproducible)
            S, Cf = self.proj_slices, cfg.conv_dim[-1]
            ld = H * Cf + H
            ops.colsum(self.proj_parts, mg("feature_projection.projection.weight").view(-1), S, H * Cf, ld)
            ops.colsum(self.proj_parts[:, H * Cf:], mg("feature_projection.projection.bias"), S, H, ld)
        self.g_proj_dn()
        ops.layernorm_bwd(self.dn, self.conv[-1].view(self.M0, -1), self.mean_f, self.rstd_f,
                          mp("feature_projection.layer_norm.weight"), self.dn, None,
                          mg("feature_projection.layer_norm.weight"), mg("feature_projection.layer_norm.bias"))
        notify("projection")
        if not st.freeze_cnn:
            if not st.cnn_runtime_frozen:         # feature_extractor.requires_grad_(False) at run time: zero gradient
                self._backward_cnn()
            notify("cnn")

    def _backward_encoder_stable(self, notify) -> None:
        """Backward of _forward_encoder_stable.  Two running gradients: self.G = d(loss)/dX[l+1] (the NORMALISED input of
        the block above: what its QKV data-gradient product wrote) and self.GR = d(loss)/dx_{l+1} (the un-normalised
        residual stream, accumulated over every block above).  Per block, top down:
            GR += LN_bwd(G)  [the LayerNorm that fed X[l+1]]      df = drop(GR)  -> FFN backward -> G = d(LN2_l output)
            GR += LN_bwd(G)  [final_layer_norm of block l]         da = drop(GR)  -> attention backward -> G = dX[l]
        A LayerDrop-skipped block only passes its re-normalisation back.  One weight-gradient launch per block (the
        grouped kernel in the 16-bit modes), LayerNorm gamma / beta folded at once; the gradient of a block's FIRST
        LayerNorm is produced by the block BELOW it (or the prologue), so the layer buckets are notified at the end."""
        cfg, st, reg = self.cfg, self.store, self.reg
        B, T, M, H = self.B, self.T, self.M, cfg.hidden_size
        mp, mg = st.mp, st.mg
        step = self._step
        heads, d = cfg.num_attention_heads, cfg.head_dim
        pa, ph = reg.attention_dropout, reg.hidden_dropout
        L = cfg.num_hidden_layers
        G, GR = self.G, self.GR
        GR.zero_()
        for l in reversed(range(L)):
            lb, gl = self.lb[l], self.g_layer[l]
            wn, bn = self._ln_params_for_input_of(l + 1)
            if l in self._skip:
                ops.layernorm_bwd(G, self._res_src[l], lb.mean2, lb.rstd2, mp(wn), G, None, mg(wn), mg(bn))
                ops.add(GR, G, GR)
                G.zero_()                       # X[l] fed a skipped block: nothing flows into it
                continue
            gs = self._gsets[l % len(self._gsets)]
            pre = f"encoder.layers.{l}."
            grouped = self.grouped
            ops.layernorm_bwd(G, lb.f, lb.mean2, lb.rstd2, mp(wn), G, None, mg(wn), mg(bn))
            ops.add(GR, G, GR)                                  # GR = d x_{l+1}
            gs["Gd"].copy_(GR)
            if ph > 0:
                ops.dropout_(gs["Gd"], ph, self._sd("ffn", l, step))
            if not grouped:
                gl["dW2"]()
                ops.colsum(gs["Gd"], mg(pre + "feed_forward.output_dense.bias"), M, H)
            gl["dh"]()
            if not grouped:
                gl["dW1"]()
                ops.colsum(gs["DH"], mg(pre + "feed_forward.intermediate_dense.bias"), M, cfg.intermediate_size)
            gl["dx1"]()                                         # G = DH @ W1 = d(LN2_l output)
            ops.layernorm_bwd(G, lb.a, lb.mean1, lb.rstd1, mp(pre + "final_layer_norm.weight"), G, None,
                              mg(pre + "final_layer_norm.weight"), mg(pre + "final_layer_norm.bias"))
            ops.add(GR, G, GR)                                  # GR = d s1 = d x_l (residual part)
            gs["Gd1"].copy_(GR)
            if ph > 0:
                ops.dropout_(gs["Gd1"], ph, self._sd("post_attn", l, step))
            if not grouped:
                gl["dWo"]()
                ops.colsum(gs["Gd1"], mg(pre + "attention.out_proj.bias"), M, H)
            gl["dctx"]()
            if self.fused:
                ops.attention_bwd(lb.qkv, lb.ctx, self.DC, lb.lse, gs["DQKV"], self.delta, B, T, heads, d, d ** -0.5,
                                  pa, self._sd("attn", l, step))
            else:
                gl["dP"]()
                ops.softmax_bwd(self.S, lb.p, self.dS, B * heads * T, T, self.Tl, pa, self._sd("attn", l, step))
                gl["dq"]()
                gl["dk"]()
                gl["dv"]()
            if grouped:
                gl["wgrad"]()
            else:
                gl["dWqkv"]()
                ops.colsum(gs["DQKV"], st.qkv(l, "g", "bias"), M, 3 * H)
            gl["dx"]()                                          # G = DQKV @ Wqkv = dX[l]
        w0, b0 = self._ln_params_for_input_of(0)
        ops.layernorm_bwd(G, self.pos, self.mean0, self.rstd0, mp(w0), G, None, mg(w0), mg(b0))
        ops.add(GR, G, GR)                                      # d x_0
        if ph > 0:
            ops.dropout_(GR, ph, self._sd("prologue", 0, step))
        G.copy_(GR)                                             # = d(hx + pos): the shared prologue backward continues
        for l in reversed(range(L)):
            notify(f"layer{l}")

    def _layer_backward_body(self, l: int, lnfold) -> None:
        """("body", l) of encoder_backward_schedule: the data-gradient chain of one transformer block."""
        cfg, st, reg = self.cfg, self.store, self.reg
        B, T, M, H = self.B, self.T, self.M, cfg.hidden_size
        mp, mg = st.mp, st.mg
        step = self._step
        heads, d = cfg.num_attention_heads, cfg.head_dim
        pa, ph = reg.attention_dropout, reg.hidden_dropout
        lb, gl = self.lb[l], self.g_layer[l]
        gs = self._gsets[l % len(self._gsets)]
        pre = f"encoder.layers.{l}."
        grouped = self.grouped
        # x2 = LN2(x1 + drop(f)):  G <- ds2 (residual path), Gd <- df = ds2 * dropmask
        ops.layernorm_bwd(self.G, lb.f, lb.mean2, lb.rstd2, mp(pre + "final_layer_norm.weight"), self.G, gs["Gd"],
                          mg(pre + "final_layer_norm.weight"), mg(pre + "final_layer_norm.bias"), ph,
                          self._sd("ffn", l, step), defer_to=lnfold)
        if not grouped:
            gl["dW2"]()
            ops.colsum(gs["Gd"], mg(pre + "feed_forward.output_dense.bias"), M, H)
        gl["dh"]()                                          # DH = (df @ W2) * gelu'(pre)   (lb.hpre holds gelu'(pre))
        if not grouped:
            gl["dW1"]()
            ops.colsum(gs["DH"], mg(pre + "feed_forward.intermediate_dense.bias"), M, cfg.intermediate_size)
        gl["dx1"]()                                         # G = DH @ W1 + G
        # x1 = LN1(x + drop(a)):  G <- ds1, Gd1 <- da
        ops.layernorm_bwd(self.G, lb.a, lb.mean1, lb.rstd1, mp(pre + "layer_norm.weight"), self.G, gs["Gd1"],
                          mg(pre + "layer_norm.weight"), mg(pre + "layer_norm.bias"), ph,
                          self._sd("post_attn", l, step), defer_to=lnfold)
        if not grouped:
            gl["dWo"]()
            ops.colsum(gs["Gd1"], mg(pre + "attention.out_proj.bias"), M, H)
        gl["dctx"]()
        if self.fused:
            ops.attention_bwd(lb.qkv, lb.ctx, self.DC, lb.lse, gs["DQKV"], self.delta, B, T, heads, d, d ** -0.5,
                              pa, self._sd("attn", l, step))
        else:
            gl["dP"]()
            ops.softmax_bwd(self.S, lb.p, self.dS, B * heads * T, T, self.Tl, pa, self._sd("attn", l, step))
            gl["dq"]()
            gl["dk"]()
            gl["dv"]()
        if not grouped:
            gl["dWqkv"]()
            ops.colsum(gs["DQKV"], st.qkv(l, "g", "bias"), M, 3 * H)

    def _backward_cnn(self) -> None:
        """Backward of the 7-layer conv feature extractor (HF:382-419) -- the reference's
        ``completely_freeze_feature_extractor: false`` ablation.  self.dn holds d(conv[6] output)."""
        cfg, st, B = self.cfg, self.store, self.Bc
        C = cfg.conv_dim
        cins = (1,) + tuple(C[:-1])
        mg = st.mg
        for i in reversed(range(1, len(C))):
            k, sd, ci, co = cfg.conv_kernel[i], cfg.conv_stride[i], cins[i], C[i]
            gout = self.dn if i == len(C) - 1 else self.dconv[i]
            gv = gout.view(-1, co)
            ops.gelu_bwd(gv, self.conv_pre[i].view(-1, co), gv)            # d(pre-activation), in place
            self.dwp.zero_()
            self.g_conv_dw[i]()
            ops.unpack_conv_grad(self.dwp, mg(f"feature_extractor.conv_layers.{i}.conv.weight"))
            self.g_conv_dx[i]()
            ops.col2im(self.col, self.dconv[i - 1], B, self.lens[i - 1], self.lens[i], ci, k, sd)
        ops.conv0_bwd(self._wav, st.mp("feature_extractor.conv_layers.0.conv.weight"), self.stats0,
                      st.mp("feature_extractor.conv_layers.0.layer_norm.weight"),
                      st.mp("feature_extractor.conv_layers.0.layer_norm.bias"), self.dconv[0], self.sums0,
                      mg("feature_extractor.conv_layers.0.conv.weight"),
                      mg("feature_extractor.conv_layers.0.layer_norm.weight"),
                      mg("feature_extractor.conv_layers.0.layer_norm.bias"), cfg.conv_kernel[0], cfg.conv_stride[0])

