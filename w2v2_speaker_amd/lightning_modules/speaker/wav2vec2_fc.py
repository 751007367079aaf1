"""Mirror of ref: src/lightning_modules/speaker/wav2vec2_fc.py (``Wav2vec2FCModule``) and of the step
semantics of ref: src/lightning_modules/speaker/speaker_recognition_module.py:148-220,322-359,462-519, on
the HIP engine.  PyTorch-Lightning is not required: the class exposes the same method names
(``compute_speaker_embedding``, ``compute_speaker_prediction``, ``forward``, ``training_step``,
``validation_step``, ``test_step``, ``generate_example_input``, ``on_train_start`` / ``on_after_backward``
freeze schedule) so a PL ``Trainer`` -- or the in-repo loop -- can drive it.

Differences by design: the loss head lives in the same flat parameter arena as the encoder (one fused Adam
launch, contiguous DDP buckets), ``training_step`` performs forward + the hand-written backward (+ the
overlapped RCCL all-reduce) itself and returns a detached loss, and ``optimizer_step`` runs the fused Adam
with the one-cycle schedule.  Only the reference's default AAM / CE configuration without hidden FC layers is
on the path (``hidden_fc_layers_out == []``, ``embedding_layer_idx == -1``)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from ...config import W2V2Config, Wav2Vec2RegularisationConfig
from ...engine import Plan
from ...evaluation.speaker.cosine_distance import CosineDistanceEvaluator, EmbeddingSample, EvaluationPair
from ...optim.schedule import OneCycle
from ...params import ParamStore
from ...trainer import SpeakerTrainer


@dataclass
class Wav2vec2FCModuleConfig:
    """Same field names (and spelling) as ref: wav2vec2_fc.py:48-98 / config/network/wav2vec2_fc.yaml."""
    wav2vec_hunggingface_id: str = "facebook/wav2vec2-base"
    reset_weights: bool = False
    wav2vec_feature_encoder_only: bool = False
    wav2vec_initially_frozen: bool = False
    num_frozen_steps: Optional[int] = 10000
    completely_freeze_feature_extractor: bool = True
    hidden_fc_layers_out: List[int] = field(default_factory=list)
    embedding_layer_idx: int = -1
    stat_pooling_type: str = "mean+std"
    test_stat_pooling_type: str = "mean+std"
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
    final_channel_mask_width: int = 5
    explicit_stat_pool_embedding_size: Optional[int] = None
    explicit_num_speakers: Optional[int] = None
    use_transformers_as_ensembles: bool = False
    num_ensembles: int = 12


@dataclass
class SpeakerClassificationDataBatch:
    """ref: src/data/modules/speaker/training_batch_speaker.py:45-75 (fields used on the hot path)."""
    batch_size: int
    keys: List[str]
    network_input: torch.Tensor      # [B, 1, N] f32 (raw, normalised waveform)
    ground_truth: torch.Tensor       # [B] int64
    side_info: Optional[Dict] = None

    def __len__(self):
        return self.batch_size

    def to(self, device) -> "SpeakerClassificationDataBatch":
        return SpeakerClassificationDataBatch(self.batch_size, self.keys, self.network_input.to(device),
                                              self.ground_truth.to(device), self.side_info)


class Wav2vec2FCModule:
    def __init__(self, cfg: Wav2vec2FCModuleConfig, num_speakers: int, loss: str = "aam",
                 aam_margin: float = 0.2, aam_scale: float = 30.0,
                 validation_pairs: Optional[List[EvaluationPair]] = None,
                 test_pairs: Optional[List[EvaluationPair]] = None, evaluator=None, *, device="cuda",
                 act_dtype: torch.dtype = torch.bfloat16, max_lr: float = 5e-5, max_steps: int = 100_000,
                 process_group=None, init_seed: int = 20211):
        if cfg.wav2vec_feature_encoder_only:
            raise NotImplementedError("Wav2vecLiteWrapperModule (CNN-only) is outside the hot path")
        if cfg.hidden_fc_layers_out or cfg.embedding_layer_idx >= 0:
            raise NotImplementedError("hidden FC layers are not on the reference's default path")
        if cfg.mask_feature_prob > 0 or cfg.final_channel_mask_prob > 0 and False:
            raise NotImplementedError("feature-axis SpecAugment (reference default mask_feature_prob=0.0)")
        assert loss in ("aam", "ce")
        self.cfg = cfg
        self.model_cfg = W2V2Config.from_huggingface_id(cfg.wav2vec_hunggingface_id)
        self.num_speakers = cfg.explicit_num_speakers or num_speakers
        self.reg = Wav2Vec2RegularisationConfig(
            activation_dropout=cfg.activation_dropout, attention_dropout=cfg.attention_dropout,
            feat_proj_dropout=cfg.feat_proj_dropout, hidden_dropout=cfg.hidden_dropout, layerdrop=cfg.layerdrop,
            mask_feature_length=cfg.mask_feature_length, mask_feature_prob=cfg.mask_feature_prob,
            mask_time_length=cfg.mask_time_length, mask_time_prob=cfg.mask_time_prob)
        H = self.model_cfg.hidden_size
        self.stat_pool_dimension = cfg.explicit_stat_pool_embedding_size or (
            2 * H if cfg.stat_pooling_type in ("mean+std", "attentive") else H)
        self.store = ParamStore(self.model_cfg, device, act_dtype, head=loss, num_speakers=self.num_speakers,
                                embed_dim=self.stat_pool_dimension,
                                freeze_cnn=cfg.completely_freeze_feature_extractor,
                                attentive_pool="attentive" in (cfg.stat_pooling_type, cfg.test_stat_pooling_type))
        self.store.init_weights(init_seed)
        self.loss, self.margin, self.scale = loss, aam_margin, aam_scale
        self.validation_pairs, self.test_pairs = validation_pairs or [], test_pairs or []
        self.evaluator = evaluator or CosineDistanceEvaluator(False, False, 0)
        self.schedule = OneCycle(max_lr=max_lr, total_steps=max_steps)
        self.process_group = process_group
        self.training = True
        self._plans: Dict[Tuple, Plan] = {}
        self._trainers: Dict[Tuple, SpeakerTrainer] = {}
        self.steps = 0
        self._is_wav2vec_frozen = False
        self.device = torch.device(device)

    # ------------------------------------------------------------------ PL-style mode switches
    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def on_train_start(self) -> None:
        # ref: wav2vec2_fc.py:339-347
        self.steps = 0
        if self.cfg.wav2vec_initially_frozen:
            self._is_wav2vec_frozen = True

    def on_after_backward(self) -> None:
        # ref: wav2vec2_fc.py:349-361 -- num_frozen_steps counts backward calls
        self.steps += 1
        if (self._is_wav2vec_frozen and self.cfg.num_frozen_steps is not None
                and self.steps >= self.cfg.num_frozen_steps):
            self._is_wav2vec_frozen = False

    # ------------------------------------------------------------------ plans
    def _plan(self, batch: int, n: int, train: bool) -> Plan:
        pooling = self.cfg.stat_pooling_type if train else self.cfg.test_stat_pooling_type
        key = (batch, n, train, pooling)
        if key not in self._plans:
            self._plans[key] = Plan(self.store, batch, n, train=train, reg=self.reg, pooling=pooling,
                                    insert_cls_token=(pooling == "first+cls"), aam_margin=self.margin,
                                    aam_scale=self.scale)
        return self._plans[key]

    @staticmethod
    def _prep_input(input_tensor: torch.Tensor) -> torch.Tensor:
        # ref: wav2vec2_fc.py:414-421 -- [BS,1,N] or [1,N] or [N] -> [BS,N]
        if len(input_tensor.shape) == 3 and input_tensor.shape[1] == 1:
            input_tensor = input_tensor[:, 0, :]
        if len(input_tensor.shape) == 1:
            input_tensor = torch.stack([input_tensor])
        return input_tensor

    # ------------------------------------------------------------------ reference surface
    def compute_speaker_embedding(self, input_tensor: torch.Tensor) -> torch.Tensor:
        x = self._prep_input(input_tensor).to(self.device, torch.float32)
        plan = self._plan(x.shape[0], x.shape[1], False)
        return plan.embed(x).clone()

    def compute_speaker_prediction(self, embedding_tensor: torch.Tensor) -> torch.Tensor:
        if self.loss == "aam":        # AAM owns the classifier weight: the "prediction" is the embedding (Q6)
            return embedding_tensor.squeeze()
        W, b = self.store.p("fc_list.0.0.weight"), self.store.p("fc_list.0.0.bias")
        from ... import ops
        B, Cn = embedding_tensor.shape[0], W.shape[0]
        out = torch.empty(B, Cn, dtype=torch.float32, device=self.device)
        ops.gemm(B, Cn, W.shape[1], embedding_tensor.float().contiguous(), W, out, lda=W.shape[1], ldb=W.shape[1],
                 ldc=Cn, epilogue=ops.EPI_BIAS, bias=b)
        return out.squeeze()

    def forward(self, input_tensor: torch.Tensor):
        embedding = self.compute_speaker_embedding(input_tensor)
        return embedding, self.compute_speaker_prediction(embedding)

    __call__ = forward

    def generate_example_input(self, include_batch_dimension: bool, batch_size: Optional[int]):
        # ref: wav2vec2_fc.py:321-337
        shape = [batch_size, 16000] if include_batch_dimension else [16000]
        return torch.rand(size=shape)

    # ------------------------------------------------------------------ steps
    def training_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0,
                      optimizer_idx: Optional[int] = None):
        """forward + backward (+ all-reduce) + fused Adam; returns {"loss", "prediction"} (device tensors)."""
        x = self._prep_input(batch.network_input).to(self.device, torch.float32)
        label = batch.ground_truth.to(self.device)
        key = (x.shape[0], x.shape[1])
        if key not in self._trainers:
            plan = self._plan(x.shape[0], x.shape[1], True)
            tr = SpeakerTrainer(self.store, plan, self.schedule, process_group=self.process_group)
            tr.step = self.steps
            self._trainers[key] = tr
        tr = self._trainers[key]
        tr.step = self.steps
        if self._is_wav2vec_frozen:
            # frozen network = eval-mode forward (PL freeze()), head-only backward + Adam
            fkey = (x.shape[0], x.shape[1], "frozen")
            if fkey not in self._plans:
                noreg = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0,
                                                     feat_proj_dropout=0.0, hidden_dropout=0.0, layerdrop=0.0,
                                                     mask_time_prob=0.0)
                self._plans[fkey] = Plan(self.store, x.shape[0], x.shape[1], train=True, reg=noreg,
                                         pooling=self.cfg.stat_pooling_type,
                                         insert_cls_token=(self.cfg.stat_pooling_type == "first+cls"),
                                         aam_margin=self.margin, aam_scale=self.scale)
            loss, pred = tr.train_step_frozen_encoder(self._plans[fkey], x, label)
        else:
            loss, pred = tr.train_step(x, label)
        self.on_after_backward()
        return {"loss": loss, "prediction": pred}

    def validation_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0):
        emb = self.compute_speaker_embedding(batch.network_input)
        return {"embedding": emb.detach().to("cpu"), "sample_id": batch.keys}

    def test_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0):
        if batch.batch_size != 1:
            raise ValueError("expecting a batch size of 1 for evaluating speaker embeddings")   # ref: :468-469
        return self.validation_step(batch, batch_idx)

    def _evaluate_embeddings(self, outputs: List[dict], pairs: List[EvaluationPair]):
        samples = []
        for o in outputs:
            for i, key in enumerate(o["sample_id"]):
                samples.append(EmbeddingSample(sample_id=key, embedding=o["embedding"][i]))
        return self.evaluator.evaluate(pairs, samples)

    def validation_epoch_end(self, outputs: List[dict]):
        return self._evaluate_embeddings(outputs, self.validation_pairs)

    def test_epoch_end(self, outputs: List[dict]):
        return self._evaluate_embeddings(outputs, self.test_pairs)

    # ------------------------------------------------------------------ checkpoints (reference key names)
    def compute_ensemble_embedding(self, input_tensor: torch.Tensor):
        """ref: wav2vec2_fc.py:440-463 -- list of ``num_ensembles`` pooled embeddings, one per hidden state of the
        last transformer layers (``use_transformers_as_ensembles``); scored by CosineDistanceEvaluator as the mean
        of the per-layer cosine scores."""
        x = input_tensor
        if x.dim() == 3 and x.shape[1] == 1:
            x = x[:, 0, :]
        if x.dim() == 1:
            x = x[None]
        key = ("ensemble", x.shape[0], x.shape[1])
        if key not in self._plans:
            self._plans[key] = Plan(self.store, x.shape[0], x.shape[1], train=False, reg=self.reg,
                                    pooling=self.cfg.stat_pooling_type, keep_hidden_states=True,
                                    insert_cls_token=(self.cfg.stat_pooling_type == "first+cls"),
                                    cls_token_constant=self.cfg.cls_token_constant
                                    if hasattr(self.cfg, "cls_token_constant") else 1.0)
        return self._plans[key].ensemble_embeddings(x.to(self.device), self.cfg.num_ensembles)

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict(sd, strict=strict, prefix_model=False)

    # ------------------------------------------------------------------ PL-format checkpoints
    def save_checkpoint(self, path: str) -> None:
        """A file ``Trainer.save_checkpoint`` / ``load_from_checkpoint`` of the reference can exchange: a pickled
        dict whose ``state_dict`` uses the reference's parameter names (``wav2vec.model.<HF name>``,
        ``loss_fn.fc_weights`` / ``fc_list.0.0.*``, ``stat_pooling.pooling_layer.*``)."""
        torch.save({"state_dict": self.state_dict(), "global_step": self.steps, "epoch": 0,
                    "pytorch-lightning_version": "1.3.8",
                    "hyper_parameters": {"num_speakers": self.num_speakers, "loss": self.loss}}, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, strict: bool = False, **kwargs) -> "Wav2vec2FCModule":
        """ref: src/main.py:272-283 (``network_class.load_from_checkpoint(path, strict=False, **kwargs)``): construct
        from kwargs, then load ``checkpoint["state_dict"]``.  Non-strict: unknown keys (e.g. BatchNorm buffers,
        ``num_batches_tracked``) are skipped, tensors whose shape differs from this module's (a head re-sized through
        ``explicit_num_speakers``) and missing ones keep their fresh initialisation."""
        module = cls(**kwargs)
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        if not strict:
            shapes = module.store.shapes
            sd = {k: v for k, v in sd.items() if k not in shapes or tuple(v.shape) == tuple(shapes[k])}
        module.load_state_dict(sd, strict=strict)
        module.steps = int(ckpt.get("global_step", 0)) if isinstance(ckpt, dict) else 0
        return module
