"""Flat parameter arena for the wav2vec2 speaker model.

All parameters live in ONE f32 device buffer, laid out in *backward order* (classification head,
encoder layers L-1..0, encoder prologue, feature projection, then the frozen CNN), so that
  * the fused Adam kernel updates every trainable parameter in one launch,
  * each data-parallel gradient bucket is a contiguous slice of the flat gradient buffer that becomes
    final exactly when backward has passed that point -> RCCL all-reduce straight on the slice, no
    copies, overlapped with the rest of backward,
  * q/k/v projection weights are adjacent, i.e. already the fused [3H, H] QKV GEMM operand.
Names are the reference's state-dict keys (``wav2vec.model.<HF name>``, ``loss_fn.fc_weights``,
``fc_list.0.0.{weight,bias}``; SURVEY 5 "Checkpoint / resume").
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .config import W2V2Config

ALIGN = 64  # elements


def hf_param_shapes(cfg: W2V2Config) -> "OrderedDict[str, Tuple[int, ...]]":
    """HF state-dict names/shapes, in the arena (= backward) order."""
    H, I, C = cfg.hidden_size, cfg.intermediate_size, cfg.conv_dim[-1]
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    shp: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for l in reversed(range(cfg.num_hidden_layers)):
        p = f"encoder.layers.{l}."
        shp[p + "final_layer_norm.weight"] = (H,)
        shp[p + "final_layer_norm.bias"] = (H,)
        shp[p + "feed_forward.output_dense.weight"] = (H, I)
        shp[p + "feed_forward.output_dense.bias"] = (H,)
        shp[p + "feed_forward.intermediate_dense.weight"] = (I, H)
        shp[p + "feed_forward.intermediate_dense.bias"] = (I,)
        shp[p + "layer_norm.weight"] = (H,)
        shp[p + "layer_norm.bias"] = (H,)
        shp[p + "attention.out_proj.weight"] = (H, H)
        shp[p + "attention.out_proj.bias"] = (H,)
        for n in ("q_proj", "k_proj", "v_proj"):           # adjacent: fused QKV operand
            shp[p + f"attention.{n}.weight"] = (H, H)
        for n in ("q_proj", "k_proj", "v_proj"):
            shp[p + f"attention.{n}.bias"] = (H,)
    shp["encoder.layer_norm.weight"] = (H,)
    shp["encoder.layer_norm.bias"] = (H,)
    shp["encoder.pos_conv_embed.conv.bias"] = (H,)
    shp["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, K)
    shp["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (H, H // G, K)
    shp["masked_spec_embed"] = (H,)
    shp["feature_projection.projection.weight"] = (H, C)
    shp["feature_projection.projection.bias"] = (H,)
    shp["feature_projection.layer_norm.weight"] = (C,)
    shp["feature_projection.layer_norm.bias"] = (C,)
    cins = (1,) + tuple(cfg.conv_dim[:-1])
    layer_norm_convs = cfg.feat_extract_norm == "layer"       # HF:275-299: a LayerNorm after every convolution
    for i in reversed(range(len(cfg.conv_dim))):
        shp[f"feature_extractor.conv_layers.{i}.conv.weight"] = (cfg.conv_dim[i], cins[i], cfg.conv_kernel[i])
        if cfg.conv_bias:
            shp[f"feature_extractor.conv_layers.{i}.conv.bias"] = (cfg.conv_dim[i],)
        if layer_norm_convs:
            shp[f"feature_extractor.conv_layers.{i}.layer_norm.weight"] = (cfg.conv_dim[i],)
            shp[f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = (cfg.conv_dim[i],)
    if not layer_norm_convs:                                  # group norm: layer 0 only (HF:302-323)
        shp["feature_extractor.conv_layers.0.layer_norm.weight"] = (cfg.conv_dim[0],)
        shp["feature_extractor.conv_layers.0.layer_norm.bias"] = (cfg.conv_dim[0],)
    return shp


W2V_PREFIX = "wav2vec.model."
# HF < 4.3x naming of the weight-norm parameters (the reference pins transformers ^4.8.2)
_WN_OLD = {"encoder.pos_conv_embed.conv.weight_g": "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
           "encoder.pos_conv_embed.conv.weight_v": "encoder.pos_conv_embed.conv.parametrizations.weight.original1"}


class ParamStore:
    def __init__(self, cfg: W2V2Config, device, act_dtype: torch.dtype = torch.bfloat16,
                 head: Optional[str] = "aam", num_speakers: int = 5994, embed_dim: Optional[int] = None,
                 freeze_cnn: bool = True, attentive_pool: bool = False, attention_channels: int = 128,
                 init_loss_scale: float = 16384.0, two_term_weights: bool = True, hidden_fc: Tuple[int, ...] = ()):
        """embed_dim: size of the pooled embedding (statistics-pooling output).  hidden_fc: output sizes of the hidden
        Linear+ReLU layers between pooling and the loss head (ref: wav2vec2_fc.py:185-228 ``hidden_fc_layers_out``);
        the CE head's Linear is fc_list.{len(hidden_fc)}.0, the AAM weight keeps input size embed_dim like the
        reference (wav2vec2_fc.py:212-224)."""
        assert act_dtype in (torch.bfloat16, torch.float16, torch.float32)
        assert head in (None, "aam", "ce", "bce")
        self.cfg, self.device, self.act_dtype = cfg, torch.device(device), act_dtype
        self.head, self.num_speakers, self.freeze_cnn = head, num_speakers, freeze_cnn
        self.embed_dim = embed_dim if embed_dim is not None else 2 * cfg.hidden_size
        shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        self.hidden_fc = tuple(int(h) for h in hidden_fc)
        dims = [self.embed_dim] + list(self.hidden_fc)
        self.head_in_dim = dims[-1]
        if head == "aam":
            if self.hidden_fc and dims[-1] != self.embed_dim:
                raise ValueError("AAM head: the reference builds the AAM weight with input_features = the pooled "
                                 f"embedding size ({self.embed_dim}); the last hidden layer must have that size")
            shapes["loss_fn.fc_weights"] = (num_speakers, self.embed_dim)
        elif head == "ce":
            n = len(self.hidden_fc)
            shapes[f"fc_list.{n}.0.weight"] = (num_speakers, dims[-1])
            shapes[f"fc_list.{n}.0.bias"] = (num_speakers,)
        if head in ("aam", "ce"):
            for i, h in enumerate(self.hidden_fc):
                shapes[f"fc_list.{i}.0.weight"] = (h, dims[i])
                shapes[f"fc_list.{i}.0.bias"] = (h,)
        if head == "bce":         # paired-input equality head (ref: wav2vec2_paired_input.py:108-110 ``self.linear``)
            self.embed_dim = embed_dim if embed_dim is not None else cfg.hidden_size
            shapes["linear.weight"] = (1, self.embed_dim)
            shapes["linear.bias"] = (1,)
        self.attentive_pool = attentive_pool
        # BatchNorm1d buffers {running_mean[A], running_var[A]} of the attentive pooling, shared by every plan
        self.asp_running = (torch.cat([torch.zeros(attention_channels), torch.ones(attention_channels)]).to(device)
                            if attentive_pool else None)
        self.asp_batches_tracked = 0
        if attentive_pool:      # pooling parameters sit with the classifier: same gradient bucket and Adam slice
            from .asp import asp_param_shapes
            shapes.update(asp_param_shapes(cfg.hidden_size, attention_channels))
        for n, s in hf_param_shapes(cfg).items():
            shapes[W2V_PREFIX + n] = s
        self.shapes = shapes
        self.offsets: Dict[str, int] = {}
        off = 0
        self.n_train = None
        for n, s in shapes.items():
            if self.n_train is None and freeze_cnn and n.startswith(W2V_PREFIX + "feature_extractor."):
                self.n_train = off
            self.offsets[n] = off
            off += (int(np.prod(s)) + ALIGN - 1) // ALIGN * ALIGN
        self.n_total = off
        if self.n_train is None:
            self.n_train = off
        # first arena element of the conv feature extractor; `cnn_runtime_frozen` is the run-time form of
        # ``feature_extractor.requires_grad_(False)`` for a store that owns CNN gradient buffers
        self.n_body = self.offsets[W2V_PREFIX + f"feature_extractor.conv_layers.{len(cfg.conv_dim) - 1}.conv.weight"]
        self.cnn_runtime_frozen = False
        dev = self.device
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        # fp16: a second plane holds the residuals fp16(W - fp16(W)) of the weights whose products run with two-term
        # weights (value and output projections of every attention block, see Plan._build_gemms / DESIGN "precision")
        self.two_term = (act_dtype == torch.float16 and two_term_weights and not os.environ.get("W2V2_NO_TWO_TERM"))
        self._lp_planes = (torch.zeros(2 if self.two_term else 1, self.n_total, dtype=act_dtype, device=dev)
                           if ops.is16(act_dtype) else None)
        self.flat_lp = self._lp_planes[0] if self._lp_planes is not None else None
        self.flat_lp_lo = self._lp_planes[1] if self.two_term else None
        self._lo_table = None
        if self.two_term:
            H = cfg.hidden_size
            rng = []
            for l in range(cfg.num_hidden_layers):
                pre = W2V_PREFIX + f"encoder.layers.{l}.attention."
                rng.append((self.offsets[pre + "v_proj.weight"], H * H))
                rng.append((self.offsets[pre + "out_proj.weight"], H * H))
            self._lo_table = torch.tensor(rng, dtype=torch.int64, device=dev)
        # fp16 activations: dynamic loss scale, device record {scale, found_inf, growth_tracker, skipped_steps}
        # (torch GradScaler semantics -- the reference trains under PL precision 16; csrc/optim.hip)
        # {scale, found_inf, growth_tracker, skipped_steps, skipped optimiser steps of the head range, ... of the body
        # range, -, -}: the last two feed Adam's bias correction (torch: a skipped step does not advance Adam's count)
        self.scaler = (torch.tensor([init_loss_scale, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
                       if act_dtype == torch.float16 else None)
        # pre-transposed bf16 copies of the 2-D weights whose data-gradient product dX = dY W is on the
        # training path (refreshed by one batched-transpose launch after every optimiser step)
        self.flat_lp_t = None
        self._t_table = None
        if self.flat_lp is not None:
            ent = []
            H = cfg.hidden_size
            for l in range(cfg.num_hidden_layers):
                pre = W2V_PREFIX + f"encoder.layers.{l}."
                o = self.offsets[pre + "attention.q_proj.weight"]
                ent.append((o, o, 3 * H, H))                     # fused QKV [3H,H] -> [H,3H]
                for n in ("attention.out_proj.weight", "feed_forward.intermediate_dense.weight",
                          "feed_forward.output_dense.weight"):
                    r, c = shapes[pre + n]
                    ent.append((self.offsets[pre + n], self.offsets[pre + n], r, c))
            n = W2V_PREFIX + "feature_projection.projection.weight"
            ent.append((self.offsets[n], self.offsets[n], shapes[n][0], shapes[n][1]))
            self.flat_lp_t = torch.zeros(self.n_total, dtype=act_dtype, device=dev)
            self._t_table = torch.tensor(ent, dtype=torch.int64, device=dev)
        self.version = 0          # bumped whenever weights change (derived packs are re-made lazily)
        self.cnn_version = 0      # bumped whenever the CNN weights change
        self.step_count = 0
        self.step_head = 0
        self.step_body = 0

    # ------------------------------------------------------------------ views
    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        s = self.shapes[name]
        o = self.offsets[name]
        return buf[o:o + int(np.prod(s))].view(*s)

    def p(self, name: str) -> torch.Tensor:
        """f32 master view."""
        return self._view(self.flat, name)

    def g(self, name: str) -> torch.Tensor:
        """f32 gradient view (trainable parameters only)."""
        if self.offsets[name] >= self.n_train:
            raise KeyError(f"{name} is frozen: no gradient")
        return self._view(self.grad, name)

    def w(self, name: str) -> torch.Tensor:
        """GEMM-operand view in the activation dtype."""
        return self._view(self.flat_lp if self.flat_lp is not None else self.flat, name)

    def wt(self, name: str) -> torch.Tensor:
        """Pre-transposed bf16 operand view [in, out] of a 2-D weight [out, in] (bf16 mode only)."""
        r, c = self.shapes[name]
        o = self.offsets[name]
        return self.flat_lp_t[o:o + r * c].view(c, r)

    def qkv_t(self, layer: int) -> torch.Tensor:
        H = self.cfg.hidden_size
        o = self.offsets[W2V_PREFIX + f"encoder.layers.{layer}.attention.q_proj.weight"]
        return self.flat_lp_t[o:o + 3 * H * H].view(H, 3 * H)

    def sync_transposed(self) -> None:
        if self.flat_lp_t is not None:
            ops.transpose_many(self.flat_lp, self.flat_lp_t, self._t_table, self._t_table.shape[0])
        if self.two_term:
            ops.weight_residual(self.flat, self.flat_lp_lo, self._lo_table)

    def w_lo(self, name: str) -> Optional[torch.Tensor]:
        """Residual-plane view of a two-term weight (None when the store has no second plane)."""
        return self._view(self.flat_lp_lo, name) if self.two_term else None

    def qkv_lo(self, layer: int) -> Optional[torch.Tensor]:
        """Residual plane of the fused [3H, H] QKV operand (only its value rows are filled)."""
        if not self.two_term:
            return None
        H = self.cfg.hidden_size
        o = self.offsets[W2V_PREFIX + f"encoder.layers.{layer}.attention.q_proj.weight"]
        return self.flat_lp_lo[o:o + 3 * H * H].view(3 * H, H)

    def is_trainable(self, name: str) -> bool:
        return self.offsets[name] < self.n_train

    def mp(self, name: str) -> torch.Tensor:
        return self.p(W2V_PREFIX + name)

    def mg(self, name: str) -> torch.Tensor:
        return self.g(W2V_PREFIX + name)

    def mw(self, name: str) -> torch.Tensor:
        return self.w(W2V_PREFIX + name)

    def qkv(self, layer: int, kind: str, which: str = "weight") -> torch.Tensor:
        """Fused [3H, H] weight (or [3H] bias) view: kind in {'p','g','w'}."""
        H = self.cfg.hidden_size
        name = W2V_PREFIX + f"encoder.layers.{layer}.attention.q_proj.{which}"
        buf = {"p": self.flat, "g": self.grad, "w": self.flat_lp if self.flat_lp is not None else self.flat}[kind]
        o = self.offsets[name]
        n = 3 * H * H if which == "weight" else 3 * H
        step = H * H if which == "weight" else H
        assert self.offsets[name.replace("q_proj", "k_proj")] == o + step, "q/k/v must be adjacent"
        assert self.offsets[name.replace("q_proj", "v_proj")] == o + 2 * step, "q/k/v must be adjacent"
        t = buf[o:o + n]
        return t.view(3 * H, H) if which == "weight" else t

    # ------------------------------------------------------------------ buckets (backward order)
    def grad_buckets(self) -> List[Tuple[str, int, int]]:
        """Contiguous gradient slices in the order backward finishes them."""
        names = list(self.shapes)
        marks: List[Tuple[str, int]] = []
        if self.head is not None or self.attentive_pool:
            marks.append(("head", 0))
        L = self.cfg.num_hidden_layers
        for l in reversed(range(L)):
            marks.append((f"layer{l}", self.offsets[W2V_PREFIX + f"encoder.layers.{l}.final_layer_norm.weight"]))
        # the encoder prologue in two buckets: encoder LayerNorm + the pos-conv bias / weight-norm pair (19 of its 20 MB)
        # are final as soon as the weight-norm backward has run, ~0.15 ms before backward ends; only the projection /
        # masked-embed / feature-LayerNorm rest (1.6 MB) is final at the very end -- the slice whose all-reduce nothing can
        # hide (profiles/r05_exposed_tail.txt)
        marks.append(("prologue", self.offsets[W2V_PREFIX + "encoder.layer_norm.weight"]))
        marks.append(("projection", self.offsets[W2V_PREFIX + "masked_spec_embed"]))
        if not self.freeze_cnn:
            marks.append(("cnn", self.offsets[W2V_PREFIX + f"feature_extractor.conv_layers.{len(self.cfg.conv_dim) - 1}.conv.weight"]))
        out = []
        for i, (n, s) in enumerate(marks):
            e = marks[i + 1][1] if i + 1 < len(marks) else self.n_train
            out.append((n, s, e))
        del names
        return out

    # ------------------------------------------------------------------ state
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, prefix_model: bool = True) -> None:
        """Accepts reference-style keys (``wav2vec.model.*``) or bare HF keys (prefix_model adds the prefix),
        and both weight-norm namings."""
        seen = set()
        buffers = self._buffers()
        for k, v in sd.items():
            if k in buffers:                               # BatchNorm running statistics of the attentive pooling
                buffers[k].copy_(torch.as_tensor(v).to(self.device, torch.float32))
                continue
            if k.endswith("num_batches_tracked") and self.asp_running is not None:
                self.asp_batches_tracked = int(v)
                continue
            k = _WN_OLD.get(k, k)
            for old, new in _WN_OLD.items():
                if k.endswith(old):
                    k = k[: -len(old)] + new
            name = k if k in self.shapes else (W2V_PREFIX + k if prefix_model and W2V_PREFIX + k in self.shapes else None)
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
            missing = [n for n in self.shapes if n not in seen]
            raise KeyError(f"missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.sync_lowp()

    def _buffers(self) -> "OrderedDict[str, torch.Tensor]":
        """Non-parameter state under the reference's key names: the BatchNorm1d buffers of the attentive pooling
        (speechbrain ``stat_pooling.pooling_layer.tdnn.norm.norm.running_{mean,var}``)."""
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        if self.asp_running is not None:
            from .asp import ASP_PREFIX
            A = self.asp_running.numel() // 2
            out[ASP_PREFIX + "tdnn.norm.norm.running_mean"] = self.asp_running[:A]
            out[ASP_PREFIX + "tdnn.norm.norm.running_var"] = self.asp_running[A:]
        return out

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict((n, self.p(n).detach().clone().cpu()) for n in self.shapes)
        for n, b in self._buffers().items():
            sd[n] = b.detach().clone().cpu()
        if self.asp_running is not None:
            from .asp import ASP_PREFIX
            sd[ASP_PREFIX + "tdnn.norm.norm.num_batches_tracked"] = torch.tensor(self.asp_batches_tracked)
        return sd

    # ------------------------------------------------------------------ reference-facing names / order
    def reference_parameter_order(self) -> List[str]:
        """Arena parameter names in the order of the reference module's ``.parameters()`` -- the indices a torch
        optimiser (and therefore a PL checkpoint's ``optimizer_states``) uses.  Registration order of
        ref: src/lightning_modules/speaker/wav2vec2_fc.py:101-228: the base class creates ``loss_fn`` first
        (speaker_recognition_module.py:62), then ``wav2vec`` (HF ``Wav2Vec2Model`` registration order, HF:1241-1262:
        masked_spec_embed, feature_extractor, feature_projection, encoder{pos_conv_embed (bias, weight-norm g, v),
        layer_norm, layers[l]{attention k/v/q/out, layer_norm, feed_forward, final_layer_norm}}), ``stat_pooling``
        (attentive pooling only) and ``fc_list``.  tests/test_host_cpu.py checks the wav2vec2 part against HF."""
        cfg, P = self.cfg, W2V_PREFIX
        names = [n for n in self.shapes if n.startswith("loss_fn.")]
        hf = ["masked_spec_embed"]
        for i in range(len(cfg.conv_dim)):
            hf.append(f"feature_extractor.conv_layers.{i}.conv.weight")
            if cfg.conv_bias:
                hf.append(f"feature_extractor.conv_layers.{i}.conv.bias")
            if i == 0 or cfg.feat_extract_norm == "layer":
                hf += [f"feature_extractor.conv_layers.{i}.layer_norm.{w}" for w in ("weight", "bias")]
        hf += [f"feature_projection.{m}.{w}" for m in ("layer_norm", "projection") for w in ("weight", "bias")]
        hf += ["encoder.pos_conv_embed.conv.bias", "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
               "encoder.pos_conv_embed.conv.parametrizations.weight.original1",
               "encoder.layer_norm.weight", "encoder.layer_norm.bias"]
        for l in range(cfg.num_hidden_layers):
            pre = f"encoder.layers.{l}."
            for m in ("attention.k_proj", "attention.v_proj", "attention.q_proj", "attention.out_proj", "layer_norm",
                      "feed_forward.intermediate_dense", "feed_forward.output_dense", "final_layer_norm"):
                hf += [pre + m + ".weight", pre + m + ".bias"]
        names += [P + n for n in hf]
        names += [n for n in self.shapes if n.startswith("stat_pooling.")]
        fc = sorted({int(n.split(".")[1]) for n in self.shapes if n.startswith("fc_list.")})
        names += [f"fc_list.{i}.0.{w}" for i in fc for w in ("weight", "bias")]
        assert sorted(names) == sorted(self.shapes), "reference_parameter_order does not cover the arena"
        return names

    @staticmethod
    def legacy_key(name: str) -> str:
        """State-dict key under the reference's own stack (torch 1.9 ``weight_norm``: ``weight_g`` / ``weight_v``;
        torch >= 2.1 parametrizations call them ``parametrizations.weight.original0/1``)."""
        for old, new in _WN_OLD.items():
            if name.endswith(new):
                return name[: -len(new)] + old
        return name

    def torch_adam_state(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, one_cycle=None) -> Dict[str, object]:
        """``torch.optim.Adam.state_dict()`` of the optimiser the reference builds over ``network.parameters()``
        (ref: src/main.py:323): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` views cut out of the flat moment
        arenas, indexed in reference parameter order; parameters that never had a gradient (frozen CNN) have no
        state entry, as in torch.  ``one_cycle`` (an ``optim.schedule.OneCycle``): the reference wraps this Adam in
        ``OneCycleLR`` (ref: src/main.py:323-335), which keeps ``initial_lr / max_lr / min_lr / base_momentum /
        max_momentum`` IN the optimiser's param group -- ``Optimizer.load_state_dict`` replaces the groups wholesale, so
        a file without them breaks ``OneCycleLR.step()`` after a resume (KeyError: 'initial_lr')."""
        order = self.reference_parameter_order()
        state = {}
        h = self.head_size()
        skipped = [int(x) for x in self.scaler[4:6].tolist()] if self.scaler is not None else [0, 0]
        step_head, step_body = self.step_head - skipped[0], self.step_body - skipped[1]     # torch's counts
        for i, n in enumerate(order):
            off = self.offsets[n]
            if self.exp_avg is None or off >= self.n_train:
                continue
            cnt = 1
            for d in self.shapes[n]:
                cnt *= d
            step = step_head if off < h else step_body
            if step <= 0:
                continue
            state[i] = {"step": step, "exp_avg": self.exp_avg[off:off + cnt].view(self.shapes[n]).detach().clone().cpu(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + cnt].view(self.shapes[n]).detach().clone().cpu()}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False}
        if one_cycle is not None:
            initial_lr = one_cycle.max_lr / one_cycle.div_factor
            group.update(initial_lr=initial_lr, max_lr=one_cycle.max_lr, min_lr=initial_lr / one_cycle.final_div_factor,
                         max_momentum=one_cycle.max_momentum, base_momentum=one_cycle.base_momentum)
        group["params"] = list(range(len(order)))
        return {"state": state, "param_groups": [group]}

    def load_torch_adam_state(self, osd: Dict[str, object]) -> None:
        """Inverse of torch_adam_state(): scatter a torch Adam state dict (reference parameter order) into the flat
        moment arenas.  Parameters without an entry keep zero moments."""
        order = self.reference_parameter_order()
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.grad)
            self.exp_avg_sq = torch.zeros_like(self.grad)
        h = self.head_size()
        sh = sb = 0
        for i, st in osd["state"].items():
            n = order[int(i)]
            off = self.offsets[n]
            if off >= self.n_train:
                continue
            ea = torch.as_tensor(st["exp_avg"]).to(self.device, torch.float32).reshape(-1)
            if tuple(torch.as_tensor(st["exp_avg"]).shape) != tuple(self.shapes[n]):
                raise ValueError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)} != {self.shapes[n]}")
            self.exp_avg[off:off + ea.numel()].copy_(ea)
            self.exp_avg_sq[off:off + ea.numel()].copy_(torch.as_tensor(st["exp_avg_sq"]).to(self.device, torch.float32).reshape(-1))
            if off < h:
                sh = max(sh, int(st["step"]))
            else:
                sb = max(sb, int(st["step"]))
        self.step_head, self.step_body = sh, sb
        self.step_count = max(sh, sb)
        if self.scaler is not None:
            self.scaler[4:6] = 0.0            # the loaded counts are torch's (skipped steps already excluded)

    # ------------------------------------------------------------------ optimiser / schedule state (resume)
    def optimizer_state(self) -> Dict[str, object]:
        """What a PL checkpoint keeps under ``optimizer_states`` (torch Adam's exp_avg / exp_avg_sq / step) plus the
        fp16 loss-scale record: a resume continues the moments, the bias correction and the scale."""
        z = lambda t: None if t is None else t.detach().clone().cpu()
        return {"exp_avg": z(self.exp_avg), "exp_avg_sq": z(self.exp_avg_sq), "step_head": self.step_head,
                "step_body": self.step_body, "loss_scaler": z(self.scaler)}

    def load_optimizer_state(self, st: Dict[str, object]) -> None:
        if st.get("exp_avg") is not None:
            ea, es = torch.as_tensor(st["exp_avg"]), torch.as_tensor(st["exp_avg_sq"])
            if ea.numel() != self.n_train:
                raise ValueError(f"optimizer state of {ea.numel()} elements does not fit this arena ({self.n_train})")
            self.exp_avg, self.exp_avg_sq = ea.to(self.device, torch.float32), es.to(self.device, torch.float32)
        self.step_head, self.step_body = int(st.get("step_head", 0)), int(st.get("step_body", 0))
        self.step_count = max(self.step_head, self.step_body)
        if self.scaler is not None and st.get("loss_scaler") is not None:
            rec = torch.as_tensor(st["loss_scaler"]).to(self.device, torch.float32).reshape(-1)
            self.scaler[:min(rec.numel(), self.scaler.numel())] = rec[:self.scaler.numel()]     # (4-float records of round 2)

    def init_weights(self, seed: int = 20211) -> None:
        """Random initialisation in the spirit of HF ``_init_weights`` (HF:1100-1140) and
        ``xavier_normal_`` for the AAM weight (ref: src/optim/loss/aam_softmax.py:38)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n, s in self.shapes.items():
            leaf = n.rsplit(".", 1)[-1]
            if n.endswith("masked_spec_embed"):
                t = torch.rand(s, generator=g)
            elif n.endswith("weight.original0"):
                t = torch.ones(s)
            elif "layer_norm" in n or ".norm.norm." in n:
                t = torch.ones(s) if leaf == "weight" else torch.zeros(s)
            elif leaf == "bias":
                t = torch.zeros(s)
            elif n == "loss_fn.fc_weights":
                t = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[0] + s[1]))
            elif n.startswith("fc_list.") and leaf == "weight":       # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
                t = (torch.rand(s, generator=g) * 2 - 1) / math.sqrt(s[1])
            elif n.endswith("original1"):
                t = torch.randn(s, generator=g) * (2.0 * math.sqrt(1.0 / (s[2] * s[1] * self.cfg.num_conv_pos_embedding_groups)))
            elif "feature_extractor" in n:
                t = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[1] * s[2]))
            else:
                t = torch.randn(s, generator=g) * 0.02
            self.p(n).copy_(t.to(self.device))
        # weight-norm: g initialised to ||v|| so that w == v (torch weight_norm semantics)
        v = self.mp("encoder.pos_conv_embed.conv.parametrizations.weight.original1")
        self.mp("encoder.pos_conv_embed.conv.parametrizations.weight.original0").copy_(
            v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.sync_lowp()

    def set_step_counts(self, step_head: int, step_body: int) -> None:
        """Host-side half of the optimiser state (Adam's bias corrections): set by a resume and by the start-up
        broadcast of the reducers, which sends rank 0's counts along with its moments."""
        self.step_head, self.step_body = int(step_head), int(step_body)
        self.step_count = max(self.step_head, self.step_body)

    def replica_state(self) -> List[torch.Tensor]:
        """Every tensor a data-parallel replica must share with rank 0 before the first step (f32 master arena incl.
        the frozen CNN and the BatchNorm buffers that live in it, Adam moments when they exist, the loss-scale record):
        what ``broadcast_parameters`` of the reducers sends.  The 16-bit operand copies are derived (sync_lowp)."""
        ts = [self.flat]
        for t in (self.exp_avg, self.exp_avg_sq, self.scaler):
            if t is not None:
                ts.append(t)
        return ts

    def sync_lowp(self) -> None:
        if self.flat_lp is not None:
            ops.cast(self.flat, self.flat_lp)
            self.sync_transposed()
        self.version += 1
        self.cnn_version += 1

    # ------------------------------------------------------------------ optimiser
    def zero_grad(self, skip_layers: Optional[Sequence[int]] = None) -> None:
        """``optimizer.zero_grad()`` of the step.  Without arguments: the whole gradient arena (400 MB on w2v2-base).
        With this step's LayerDrop decisions (``skip_layers``, possibly empty) and the grouped weight-gradient path
        (16-bit modes): only what the backward ACCUMULATES into -- LayerNorm gamma / beta, pos-conv bias and weight-norm
        pair, masked_spec_embed, the head's small tensors -- plus the whole slice of every skipped layer.  The Linear
        weights / biases of the encoder (99 % of the arena) are WRITTEN by w2v2_wgrad_grouped, the AAM weight by the
        head's normalisation backward, so zeroing them first is wasted HBM traffic."""
        if skip_layers is None or self.flat_lp_t is None or not self.freeze_cnn:
            self.grad.zero_()
            return
        if getattr(self, "_zero_tables", None) is None:
            self._zero_tables = self._build_zero_tables()
        small, per_layer = self._zero_tables
        if small.shape[0]:
            ops.zero_ranges(self.grad, small, blocks_per_range=16)
        for l in skip_layers:
            ops.zero_ranges(self.grad, per_layer[l], blocks_per_range=512)

    def _build_zero_tables(self):
        """(table of the accumulated tensors' (offset, count) ranges, one single-row table per transformer layer)."""
        P, H = W2V_PREFIX, self.cfg.hidden_size
        written = set()
        for l in range(self.cfg.num_hidden_layers):
            pre = P + f"encoder.layers.{l}."
            for m in ("attention.q_proj", "attention.k_proj", "attention.v_proj", "attention.out_proj",
                      "feed_forward.intermediate_dense", "feed_forward.output_dense"):
                written |= {pre + m + ".weight", pre + m + ".bias"}
        # (the projection's gradient is the fixed-order SUM of per-slice partials added into a zeroed target: not listed)
        if self.head == "aam":
            written.add("loss_fn.fc_weights")
        # the weight-norm backward WRITES both halves of the pos-conv weight pair (csrc/posconv.hip wn_bwd_*): 19 MB that
        # round 4 zeroed first, on 4 workgroups (39 us)
        written |= {P + "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
                    P + "encoder.pos_conv_embed.conv.parametrizations.weight.original1"}
        rng = []
        for n, off in sorted(self.offsets.items(), key=lambda kv: kv[1]):
            if off >= self.n_train or n in written:
                continue
            cnt = int(np.prod(self.shapes[n]))
            if rng and rng[-1][0] + rng[-1][1] == off:
                rng[-1][1] += cnt
            else:
                rng.append([off, cnt])
        small = torch.tensor(rng, dtype=torch.int64, device=self.device).reshape(-1, 2)
        buckets = {n: (s, e) for n, s, e in self.grad_buckets()}
        per_layer = [torch.tensor([[buckets[f"layer{l}"][0], buckets[f"layer{l}"][1] - buckets[f"layer{l}"][0]]],
                                  dtype=torch.int64, device=self.device) for l in range(self.cfg.num_hidden_layers)]
        return small, per_layer

    def head_size(self) -> int:
        """Number of leading arena elements that belong to the classification head."""
        return self.grad_buckets()[0][2] if (self.head is not None or self.attentive_pool) else 0

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                  grad_scale: float = 1.0, head_only: bool = False) -> None:
        """head_only: the wav2vec2 network is frozen (ref: wav2vec2_fc.py:339-361 ``wav2vec_initially_frozen``),
        only the head slice of the arena is updated.

        torch.optim.Adam keeps a step count PER PARAMETER that only advances when the parameter has a
        gradient: parameters frozen for the first steps start their bias correction at 1 when they
        unfreeze.  Two counters (head / rest of the arena) reproduce that."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.grad)
            self.exp_avg_sq = torch.zeros_like(self.grad)
        h = self.head_size()
        self.step_head += 1
        if not head_only:
            self.step_body += 1
        self.step_count = max(self.step_head, self.step_body)
        a = (self.flat, self.grad, self.exp_avg, self.exp_avg_sq)
        lp, sc = self.flat_lp, self.scaler
        n_train = min(self.n_train, self.n_body) if self.cnn_runtime_frozen else self.n_train
        if sc is not None:
            # found_inf (torch GradScaler.unscale_): an overflow of ANY fp16 activation gradient (the only 16-bit
            # tensors of the backward; weight gradients are f32 sums of finite products) propagates down the chain into
            # the LAST bucket backward writes, so scanning that bucket (1.4 M of 99 M elements) decides for the step.
            # Head-only steps and steps with a trainable CNN scan their whole slice.
            if head_only:
                ops.grad_scaler_check(self.grad, h, sc)
            elif n_train > self.n_body or os.environ.get("W2V2_SCALER_FULL_SCAN"):      # (debug: scan the whole arena)
                ops.grad_scaler_check(self.grad, n_train, sc)
            else:
                lo = self.offsets[W2V_PREFIX + "encoder.layer_norm.weight"]
                ops.grad_scaler_check(self.grad[lo:], n_train - lo, sc)
        sh, sb = (4, 5) if sc is not None else (0, 0)         # record slots of the skipped-step counts (head / body)
        if head_only:
            ops.adam_step(*a, lp, h, lr, beta1, beta2, eps, self.step_head, grad_scale, sc, sh)
        elif self.step_head == self.step_body or h == 0:
            ops.adam_step(*a, lp, n_train, lr, beta1, beta2, eps, self.step_body, grad_scale, sc, sb)
        else:
            ops.adam_step(*a, lp, h, lr, beta1, beta2, eps, self.step_head, grad_scale, sc, sh)
            ops.adam_step(*(t[h:] for t in a), lp[h:] if lp is not None else None, n_train - h, lr, beta1,
                          beta2, eps, self.step_body, grad_scale, sc, sb)
        if sc is not None:
            ops.grad_scaler_update(sc, skipped_ranges=1 if head_only else 3)
        self.sync_transposed()
        self.version += 1
        if not self.freeze_cnn:
            self.cnn_version += 1
