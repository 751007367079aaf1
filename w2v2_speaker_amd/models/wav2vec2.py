"""Mirror of ref: src/models/wav2vec2.py -- ``Wav2Vec2WrapperModule`` over the HIP engine.

Same constructor arguments, same ``forward(wav [B,N]) -> [B, num_features, num_frames]`` contract
(ref: src/models/wav2vec2.py:97-146), same state-dict keys (``model.<HF name>``).  The arithmetic is
engine.Plan (hand-written HIP kernels); torch autograd only sees one opaque Function whose backward is
the engine's hand-written backward."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from ..config import W2V2Config, Wav2Vec2RegularisationConfig  # noqa: F401  (re-exported like the reference)
from ..engine import Plan
from ..params import ParamStore, W2V_PREFIX


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
                 act_dtype: torch.dtype = torch.bfloat16, init_seed: int = 20211):
        super().__init__()
        self.cfg = W2V2Config.from_huggingface_id(wav2vec2_huggingface_id)   # "base" / "large" substring rule
        self.num_features = self.cfg.hidden_size
        self.insert_cls_token = insert_clc_token
        self.cls_token_constant = cls_token_constant
        self.reg_cfg = reg_cfg if reg_cfg is not None else Wav2Vec2RegularisationConfig()
        self.store = store if store is not None else ParamStore(self.cfg, device, act_dtype, head=None)
        if store is None:
            # no network: "from_pretrained" weights must be loaded with load_state_dict(); reset_weights and the
            # offline default both give a fresh random initialisation (ref: src/util.py:214-226 semantics)
            self.store.init_weights(init_seed)
        self._plans: Dict[Tuple[int, int, bool], Plan] = {}
        self._anchor = torch.nn.Parameter(torch.zeros((), device=self.store.device))
        self._step = 0

    @property
    def num_embedding_features(self):
        return self.num_features

    def _plan(self, batch: int, n_samples: int, train: bool) -> Plan:
        key = (batch, n_samples, train)
        if key not in self._plans:
            self._plans[key] = Plan(self.store, batch, n_samples, train=train, reg=self.reg_cfg,
                                    insert_cls_token=self.insert_cls_token,
                                    cls_token_constant=self.cls_token_constant)
        return self._plans[key]

    def _publish_grads(self) -> None:
        pass            # gradients live in store.grad (flat arena); optimisers use ParamStore.adam_step

    def forward(self, wav_input: torch.Tensor, mask_time_indices: Optional[torch.Tensor] = None,
                skip_layers=()) -> torch.Tensor:
        # wav_input has shape [BATCH_SIZE, NUM_SAMPLES]
        out = _EncoderFn.apply(wav_input.to(self.store.device, torch.float32), self._anchor, self,
                               mask_time_indices, tuple(skip_layers))
        # return an embedding with shape [BATCH_SIZE, NUM_FEATURES, NUM_FRAMES]
        return out.transpose(1, 2)

    # state dict with the reference's key names: model.<HF name>
    def state_dict(self, *args, prefix: str = "", **kw):
        return {prefix + "model." + k[len(W2V_PREFIX):]: v for k, v in self.store.state_dict().items()
                if k.startswith(W2V_PREFIX)}

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict({(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()},
                                   strict=False if not strict else False)
        for p in self._plans.values():
            p._pack_version = -1
