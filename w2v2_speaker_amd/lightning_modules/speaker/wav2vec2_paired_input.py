"""Mirror of ``src/lightning_modules/speaker/wav2vec2_paired_input.py`` (Wav2vec2PairedSpeakerModuleConfig :27-67,
Wav2vec2PairedSpeakerModule :70-207) + ``paired_speaker_recognition_module.py:60-140`` on the HIP path
(engine.Plan(paired=True), heads.BceHead)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ...config import W2V2Config, Wav2Vec2RegularisationConfig
from ...engine import Plan
from ...optim.schedule import OneCycle
from ...params import ParamStore
from ...trainer import SpeakerTrainer


@dataclass
class Wav2vec2PairedSpeakerModuleConfig:
    """ref: wav2vec2_paired_input.py:27-67 (same field names, incl. the reference's spelling)."""
    wav2vec_hunggingface_id: str = "facebook/wav2vec2-base"
    reset_weights: bool = True
    wav2vec_initially_frozen: bool = False
    num_frozen_steps: Optional[int] = None
    completely_freeze_feature_extractor: bool = True
    completely_freeze_feature_projector: bool = False
    cls_token_constant: float = 1
    sep_token_constant: float = -1
    activation_dropout: float = 0.0
    attention_dropout: float = 0.1
    feat_proj_dropout: float = 0.1
    hidden_dropout: float = 0.1
    layerdrop: float = 0.05
    mask_feature_length: int = 10
    mask_feature_prob: float = 0.0
    mask_time_length: int = 10
    mask_time_prob: float = 0.05
    final_channel_mask_prob: float = 0.0
    final_channel_mask_width: int = 0


@dataclass
class PairedSpeakerClassificationDataBatch:
    """ref: src/data/modules/speaker/training_batch_speaker.py (paired batch): two waveforms per pair + {0,1} label."""
    batch_size: int
    primary_keys: List[str]
    primary_network_input: torch.Tensor
    secondary_keys: List[str]
    secondary_network_input: torch.Tensor
    ground_truth: torch.Tensor


class Wav2vec2PairedSpeakerModule:
    def __init__(self, hyperparameters_to_save, cfg: Wav2vec2PairedSpeakerModuleConfig,
                 loss_fn_constructor: Optional[Callable[[], object]] = None, *, device="cuda",
                 act_dtype: torch.dtype = torch.float16, max_lr: float = 5e-5, max_steps: int = 100_000,
                 process_group=None, init_seed: int = 20211):
        """Positional arguments = ref: wav2vec2_paired_input.py:65-71.  ``loss_fn_constructor`` must build the
        reference's ``BinaryCrossEntropyLoss`` (src/optim/loss/binary_cross_entropy.py; the only loss this module is
        configured with, config/optim/loss/binary_cross_entropy.yaml) -- it is called once and checked; the arithmetic
        runs in w2v2_bce_head_fwd_bwd.  Default precision: fp16 operands under the dynamic loss scale, like
        Wav2vec2FCModule (the reference's ``precision: 16``)."""
        if cfg.wav2vec_initially_frozen or cfg.completely_freeze_feature_projector:
            raise NotImplementedError("initially-frozen network / frozen projector for the paired module")
        if loss_fn_constructor is not None:
            from ...optim.loss import BinaryCrossEntropyLoss
            loss_fn = loss_fn_constructor()
            if not isinstance(loss_fn, BinaryCrossEntropyLoss):
                raise NotImplementedError(f"loss {type(loss_fn).__name__}: the paired module trains with "
                                          "BinaryCrossEntropyLoss")
        self.hyperparameters_to_save = hyperparameters_to_save
        self.cfg = cfg
        self.model_cfg = W2V2Config.from_huggingface_id(cfg.wav2vec_hunggingface_id)
        self.reg = Wav2Vec2RegularisationConfig(
            activation_dropout=cfg.activation_dropout, attention_dropout=cfg.attention_dropout,
            feat_proj_dropout=cfg.feat_proj_dropout, hidden_dropout=cfg.hidden_dropout, layerdrop=cfg.layerdrop,
            mask_feature_length=cfg.mask_feature_length, mask_feature_prob=cfg.mask_feature_prob,
            mask_time_length=cfg.mask_time_length, mask_time_prob=cfg.mask_time_prob)
        self.store = ParamStore(self.model_cfg, device, act_dtype, head="bce",
                                freeze_cnn=cfg.completely_freeze_feature_extractor)
        self.store.init_weights(init_seed)
        self.schedule = OneCycle(max_lr=max_lr, total_steps=max_steps)
        self.process_group = process_group
        self.device = torch.device(device)
        self._plans: Dict[Tuple, Plan] = {}
        self._trainers: Dict[Tuple, SpeakerTrainer] = {}
        self.steps = 0

    def _get_wav2vec2_embedding_size(self):
        return self.model_cfg.hidden_size                   # ref :112-118 (768 / 1024)

    def generate_example_input(self, include_batch_dimension: bool, batch_size: Optional[int] = None):
        shape = [batch_size, 16000] if include_batch_dimension else [16000]
        return torch.rand(size=shape), torch.rand(size=shape)

    def _plan(self, batch: int, n: int, train: bool) -> Plan:
        key = (batch, n, train)
        if key not in self._plans:
            self._plans[key] = Plan(self.store, batch, n, train=train, reg=self.reg, pooling="first", paired=True,
                                    cls_token_constant=self.cfg.cls_token_constant,
                                    sep_token_constant=self.cfg.sep_token_constant)
        return self._plans[key]

    @staticmethod
    def _stack(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        sq = lambda x: x[:, 0, :] if x.dim() == 3 else (x[None] if x.dim() == 1 else x)
        a, b = sq(a), sq(b)
        assert a.shape == b.shape                          # ref :166-168
        return torch.cat([a, b], dim=0)

    def compute_speaker_equality(self, wav_tensor: torch.Tensor, other_wav_tensor: torch.Tensor) -> torch.Tensor:
        """ref :163-207 -> equality logits [B, 1] (eval-mode forward)."""
        wav = self._stack(wav_tensor, other_wav_tensor).to(self.device, torch.float32)
        plan = self._plan(wav.shape[0] // 2, wav.shape[1], False)
        emb = plan.embed(wav)
        from ... import ops
        B, H = emb.shape
        out = torch.zeros(B, 4, dtype=torch.float32, device=self.device)          # ldc padded to 4
        ops.gemm(B, 1, H, emb, self.store.p("linear.weight"), out, lda=H, ldb=H, ldc=4, epilogue=ops.EPI_BIAS,
                 bias=self.store.p("linear.bias"))
        return out[:, :1]

    def forward(self, input_tensor: torch.Tensor, other_input_tensor: torch.Tensor):
        return self.compute_speaker_equality(input_tensor, other_input_tensor)

    __call__ = forward

    def training_step(self, batch: PairedSpeakerClassificationDataBatch, batch_idx: int = 0,
                      optimized_idx: Optional[int] = None):
        """ref: paired_speaker_recognition_module.py:68-90: forward, BCE, backward (+ all-reduce), fused Adam."""
        wav = self._stack(batch.primary_network_input, batch.secondary_network_input).to(self.device, torch.float32)
        label = batch.ground_truth.to(self.device).to(torch.int64)
        key = (wav.shape[0] // 2, wav.shape[1])
        if key not in self._trainers:
            self._trainers[key] = SpeakerTrainer(self.store, self._plan(key[0], key[1], True), self.schedule,
                                                 process_group=self.process_group)
        tr = self._trainers[key]
        tr.step = self.steps
        loss, pred = tr.train_step(wav, label)
        self.steps += 1
        return {"loss": loss, "prediction": pred}

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict(sd, strict=strict, prefix_model=False)
