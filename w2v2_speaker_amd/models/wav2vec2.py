"""Mirror of ref: src/models/wav2vec2.py -- ``Wav2Vec2WrapperModule`` over the HIP engine.

Same constructor arguments, same ``forward(wav [B,N]) -> [B, num_features, num_frames]`` contract
(ref: src/models/wav2vec2.py:97-146), same state-dict keys (``model.<HF name>``).  The arithmetic is
engine.Plan (hand-written HIP kernels); torch autograd only sees one opaque Function whose backward is
the engine's hand-written backward."""
from __future__ import annotations

import warnings
from collections import OrderedDict
from typing import Optional, Tuple

import numpy as np
import torch

from ..config import W2V2Config, Wav2Vec2RegularisationConfig  # noqa: F401  (re-exported like the reference)
from ..engine import Plan
from ..params import ParamStore, W2V_PREFIX
from ..spec_augment import compute_mask_indices
from .handles import ModelHandle

MAX_PLANS = 8


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, anchor, module, mask, skip_layers):
        plan = module._plan(wav.shape[0], wav.shape[-1], module.training)
        module._step += 1
        out = plan.forward(wav, mask, skip_layers, module._step)
        ctx.plan = plan
        ctx.module = module
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        if not plan.train:
            raise RuntimeError("backward through Wav2Vec2WrapperModule needs module.train()")
        plan.backward(dhidden=dout.to(plan.adt).contiguous())
        ctx.module._publish_grads()
        return None, torch.zeros((), device=dout.device), None, None, None


class Wav2Vec2WrapperModule(torch.nn.Module):
    def __init__(self, wav2vec2_huggingface_id: str, reset_weights: bool,
                 reg_cfg: Optional[Wav2Vec2RegularisationConfig] = None, insert_clc_token: bool = False,
                 cls_token_constant: float = 1, *, store: Optional[ParamStore] = None, device="cuda",
                 act_dtype: torch.dtype = torch.float16, init_seed: int = 20211, pretrained_state_dict=None,
                 regularisation_seed: int = 1234):
        """Positional arguments = ref: src/models/wav2vec2.py:97-104.  ``pretrained_state_dict`` (path or dict of HF
        ``facebook/wav2vec2-*`` weights) replaces ``from_pretrained`` (no network here).  Gradients of a backward pass
        through this module are ACCUMULATED in ``store.grad`` (the flat arena): call ``store.zero_grad()`` per step
        and step with ``store.adam_step``."""
        super().__init__()
        self.cfg = W2V2Config.from_huggingface_id(wav2vec2_huggingface_id)   # "base" / "large" substring rule
        self.num_features = self.cfg.hidden_size
        self.insert_cls_token = insert_clc_token
        self.cls_token_constant = cls_token_constant
        self.reg_cfg = reg_cfg if reg_cfg is not None else Wav2Vec2RegularisationConfig()
        self.store = store if store is not None else ParamStore(self.cfg, device, act_dtype, head=None)
        if store is None:
            self.store.init_weights(init_seed)       # ref: src/util.py:214-226 reset_model (fresh initialisation)
            if pretrained_state_dict is not None:
                sd = (torch.load(pretrained_state_dict, map_location="cpu", weights_only=False)
                      if isinstance(pretrained_state_dict, str) else pretrained_state_dict)
                self.store.load_state_dict(dict(sd), strict=False, prefix_model=True)
            elif not reset_weights:
                warnings.warn(f"Wav2Vec2WrapperModule: reset_weights=False asks for the pretrained "
                              f"{wav2vec2_huggingface_id!r} weights but no pretrained_state_dict was given (there is "
                              "no network access): the model starts from a RANDOM initialisation", stacklevel=2)
        self._plans: "OrderedDict[Tuple[int, int, bool], Plan]" = OrderedDict()
        self._anchor = torch.nn.Parameter(torch.zeros((), device=self.store.device))
        self._step = 0
        self._ld_rng = np.random.RandomState(regularisation_seed)
        self._mask_rng = np.random.RandomState(regularisation_seed + 1)
        # ``wrapper.model.{feature_extractor,feature_projection,encoder}`` of the reference's callers
        self.model = ModelHandle(self.store)

    @property
    def num_embedding_features(self):
        return self.num_features

    def _plan(self, batch: int, n_samples: int, train: bool) -> Plan:
        key = (batch, n_samples, train)
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        self._plans[key] = Plan(self.store, batch, n_samples, train=train, reg=self.reg_cfg,
                                insert_cls_token=self.insert_cls_token, cls_token_constant=self.cls_token_constant)
        while len(self._plans) > MAX_PLANS:             # bounded: one static plan per (batch, length, mode)
            self._plans.popitem(last=False)
        return self._plans[key]

    def _publish_grads(self) -> None:
        pass            # gradients live in store.grad (flat arena); optimisers use ParamStore.adam_step

    def forward(self, wav_input: torch.Tensor, mask_time_indices: Optional[torch.Tensor] = None,
                skip_layers=None) -> torch.Tensor:
        """wav_input [BATCH_SIZE, NUM_SAMPLES].  In train mode the SpecAugment time mask (HF:1272-1316) and the
        LayerDrop decisions (HF:698-709) are sampled from ``reg_cfg`` like the HF model does, unless given."""
        if self.training:
            reg, B = self.reg_cfg, wav_input.shape[0]
            if skip_layers is None:
                u = self._ld_rng.rand(self.cfg.num_hidden_layers)
                skip_layers = tuple(int(i) for i in np.nonzero(u < reg.layerdrop)[0]) if reg.layerdrop > 0 else ()
            if mask_time_indices is None and reg.mask_time_prob > 0 and not self.insert_cls_token:
                m = compute_mask_indices((B, self.cfg.num_frames(wav_input.shape[-1])), reg.mask_time_prob,
                                         reg.mask_time_length, self.cfg.mask_time_min_masks, rng=self._mask_rng)
                mask_time_indices = torch.from_numpy(m.astype(np.uint8)).to(self.store.device)
        skip_layers = tuple(skip_layers or ())
        out = _EncoderFn.apply(wav_input.to(self.store.device, torch.float32), self._anchor, self,
                               mask_time_indices, tuple(skip_layers))
        # return an embedding with shape [BATCH_SIZE, NUM_FEATURES, NUM_FRAMES]
        return out.transpose(1, 2)

    # state dict with the reference's key names: model.<HF name>
    def state_dict(self, *args, prefix: str = "", **kw):
        return {prefix + "model." + k[len(W2V_PREFIX):]: v for k, v in self.store.state_dict().items()
                if k.startswith(W2V_PREFIX)}

    def load_state_dict(self, sd, strict: bool = True):
        """strict: every encoder parameter must be present and no unknown key may appear (keys outside the wav2vec2
        network -- a loss head that lives in the same store -- are not this module's and never required)."""
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        if strict:
            own = {n[len(W2V_PREFIX):] for n in self.store.shapes if n.startswith(W2V_PREFIX)}
            alias = {"encoder.pos_conv_embed.conv.weight_g": "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
                     "encoder.pos_conv_embed.conv.weight_v": "encoder.pos_conv_embed.conv.parametrizations.weight.original1"}
            got = {alias.get(k, k) for k in sd}
            if own - got:
                raise KeyError(f"missing keys: {sorted(own - got)[:5]}")
            if got - own:
                raise KeyError(f"unexpected keys: {sorted(got - own)[:5]}")
        self.store.load_state_dict(sd, strict=False)
        for p in self._plans.values():
            p._pack_version = -1


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, anchor, module):
        plan = module._plan(wav.shape[0], wav.shape[-1], module.training and not module.store.freeze_cnn)
        ctx.plan = plan
        return plan.conv_features(wav).clone()

    @staticmethod
    def backward(ctx, dout):
        ctx.plan.conv_backward(dout)
        return None, torch.zeros((), device=dout.device), None


class Wav2vecLiteWrapperModule(torch.nn.Module):
    """Mirror of ref: src/models/wav2vec2.py:149-169 -- the wav2vec2 CONV FEATURE EXTRACTOR only
    (``wav2vec_feature_encoder_only``): ``forward(wav [B,N]) -> [B, 512, num_frames]``, on the engine's conv kernels
    (layer 0 fused conv + GroupNorm + GELU, layers 1-6 implicit-GEMM convs).  Gradients of a backward pass are
    accumulated in ``store.grad`` like in Wav2Vec2WrapperModule (the store must be built with ``freeze_cnn=False`` to
    train the CNN; with the default frozen store the module is forward-only)."""
    num_features = 512

    def __init__(self, wav2vec2_huggingface_id: str, reset_weights: bool, *, store: Optional[ParamStore] = None,
                 device="cuda", act_dtype: torch.dtype = torch.float16, init_seed: int = 20211, pretrained_state_dict=None,
                 freeze_cnn: bool = False):
        super().__init__()
        self.cfg = W2V2Config.from_huggingface_id(wav2vec2_huggingface_id)
        self.num_features = self.cfg.conv_dim[-1]
        self.store = store if store is not None else ParamStore(self.cfg, device, act_dtype, head=None,
                                                                freeze_cnn=freeze_cnn)
        if store is None:
            self.store.init_weights(init_seed)
            if pretrained_state_dict is not None:
                sd = (torch.load(pretrained_state_dict, map_location="cpu", weights_only=False)
                      if isinstance(pretrained_state_dict, str) else pretrained_state_dict)
                self.store.load_state_dict(dict(sd), strict=False, prefix_model=True)
            elif not reset_weights:
                warnings.warn("Wav2vecLiteWrapperModule: reset_weights=False asks for pretrained weights but no "
                              "pretrained_state_dict was given (no network access): RANDOM initialisation", stacklevel=2)
        self._plans: "OrderedDict[Tuple[int, int, bool], Plan]" = OrderedDict()
        self._anchor = torch.nn.Parameter(torch.zeros((), device=self.store.device))
        self.model = ModelHandle(self.store)

    @property
    def num_embedding_features(self):
        return self.num_features

    def _plan(self, batch: int, n_samples: int, train: bool) -> Plan:
        key = (batch, n_samples, train)
        if key not in self._plans:
            self._plans[key] = Plan(self.store, batch, n_samples, train=train, reg=Wav2Vec2RegularisationConfig())
            while len(self._plans) > MAX_PLANS:
                self._plans.popitem(last=False)
        self._plans.move_to_end(key)
        return self._plans[key]

    def forward(self, wav_input: torch.Tensor) -> torch.Tensor:
        # wav_input has shape [BATCH_SIZE, NUM_SAMPLES]
        feat = _ConvFn.apply(wav_input.to(self.store.device, torch.float32), self._anchor, self)
        # return an embedding with shape [BATCH_SIZE, NUM_FEATURES, NUM_FRAMES]
        return feat.transpose(1, 2)
