"""Mirror of ref: src/lightning_modules/speaker/wav2vec2_fc.py (``Wav2vec2FCModule``) and of the step
semantics of ref: src/lightning_modules/speaker/speaker_recognition_module.py:148-220,296-307,322-359,462-519, on
the HIP engine.  PyTorch-Lightning is not required: the class has the reference's constructor
(``hyperparameters_to_save, cfg, num_speakers, loss_fn_constructor, validation_pairs, test_pairs, evaluator`` --
what ref: src/main.py:223-285 passes) and the same method names (``compute_speaker_embedding``,
``compute_speaker_prediction``, ``forward``, ``training_step``, ``validation_step``, ``test_step``,
``generate_example_input``, ``on_train_start`` / ``on_after_backward`` freeze schedule, ``wav2vec.model.*`` handles)
so a PL ``Trainer`` -- or the in-repo loop -- can drive it.

Differences by design: the loss head lives in the same flat parameter arena as the encoder (one fused Adam
launch, contiguous DDP buckets), ``training_step`` performs forward + the hand-written backward (+ the
overlapped RCCL all-reduce) + fused Adam itself and returns a detached loss.  ``loss_fn_constructor`` is called once
and read for its type and hyper-parameters (AAM ``margin`` / ``scale``): the arithmetic of the loss runs in the
engine's head kernels on the arena's weight, like the reference re-creates its AAM module with the right sizes
(ref: wav2vec2_fc.py:212-224)."""
from __future__ import annotations

import warnings
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ... import ops
from ...config import W2V2Config, Wav2Vec2RegularisationConfig
from ...engine import Plan
from ...evaluation.speaker.cosine_distance import CosineDistanceEvaluator, EmbeddingSample, EvaluationPair
from ...models.handles import ModelHandle
from ...optim.loss import AngularAdditiveMarginSoftMaxLoss, CrossEntropyLoss
from ...optim.schedule import OneCycle
from ...params import ParamStore
from ...trainer import SpeakerTrainer

from ...params import _WN_OLD as _WN_LEGACY


def _load_checkpoint_file(path: str, trust_pickle: bool):
    """Checkpoints and pretrained weight files hold tensors, ints, strings and containers of them: load with
    ``weights_only=True`` (no arbitrary unpickling of a downloaded file).  A third-party checkpoint that pickles other
    objects (e.g. an OmegaConf ``hyper_parameters`` node) needs the explicit opt-in ``trust_checkpoint_pickle=True``."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        if not trust_pickle:
            raise
        return torch.load(path, map_location="cpu", weights_only=False)


MAX_PLANS = 8      # static plans kept per module (LRU): evaluation over variable-length utterances builds one per length


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


class _Wav2vecHandle:
    """``module.wav2vec`` of the reference (a Wav2Vec2WrapperModule): ``.model.{feature_extractor,
    feature_projection,encoder}``, ``.num_features``, PL's ``freeze()`` / ``unfreeze()`` (ref: wav2vec2_fc.py:339-361)."""

    def __init__(self, owner: "Wav2vec2FCModule"):
        self._owner = owner
        self.model = ModelHandle(owner.store, on_body_grad=owner._set_body_trainable)
        self.num_features = owner.model_cfg.hidden_size

    @property
    def num_embedding_features(self):
        return self.num_features

    def freeze(self) -> None:          # PL LightningModule.freeze(): requires_grad False + eval mode
        self._owner._is_wav2vec_frozen = True

    def unfreeze(self) -> None:
        self._owner._is_wav2vec_frozen = False


def _pool_width(pooling: str) -> int:
    """ref: wav2vec2_fc.py:290-319 _determine_stat_pool_embedding_size."""
    p = pooling.lower()
    if p in ("mean", "first", "first+cls", "last", "middle", "random", "max", "none"):
        return 1
    if p in ("mean+std", "attentive"):
        return 2
    if p == "quantile":
        return 5
    raise ValueError(f"unknown value for stat_pooling_type={pooling!r}")


class Wav2vec2FCModule(torch.nn.Module):
    """A ``torch.nn.Module`` (the reference's is a LightningModule): ``parameters()`` / ``named_parameters()`` yield one
    ``nn.Parameter`` per reference parameter, in the reference's order and under its names, each a VIEW of the flat f32
    arena (``.grad`` = the matching view of the flat gradient buffer, scaled by the loss scale in the fp16 mode);
    ``state_dict()`` / ``load_state_dict()`` speak the reference's keys; ``train()`` / ``eval()`` are nn.Module's.
    ``automatic_optimization = False``: ``training_step`` runs forward, the hand-written backward, the gradient
    all-reduce and the fused Adam itself (PL's manual-optimisation contract), an optimiser built over ``parameters()``
    is therefore not stepped by this module.  ``.to()`` / ``.half()`` / ``.cuda()`` are no-ops: the engine is bound to
    the device and activation dtype it was constructed with."""
    automatic_optimization = False

    def __init__(self, hyperparameters_to_save, cfg: Wav2vec2FCModuleConfig, num_speakers: int,
                 loss_fn_constructor: Callable[[], object], validation_pairs: Optional[List[EvaluationPair]] = None,
                 test_pairs: Optional[List[EvaluationPair]] = None, evaluator=None, *, device="cuda",
                 act_dtype: torch.dtype = torch.float16, max_lr: float = 5e-5, max_steps: int = 100_000,
                 process_group=None, init_seed: int = 20211, pretrained_state_dict=None):
        """Positional arguments = ref: wav2vec2_fc.py:101-111.  Keyword-only extras: the device / activation dtype of
        the engine, the one-cycle schedule the reference takes from ``cfg.optim`` (src/main.py:323-335), and
        ``pretrained_state_dict`` (a path or a dict with HF ``facebook/wav2vec2-*`` weights: there is no network
        here for ``from_pretrained``)."""
        super().__init__()
        self.hyperparameters_to_save = hyperparameters_to_save
        self.cfg = cfg
        if cfg.wav2vec_feature_encoder_only:
            # ref: :114-128 swaps in Wav2vecLiteWrapperModule (CNN only).  The wrapper itself is mirrored
            # (models.wav2vec2.Wav2vecLiteWrapperModule, forward + conv backward); the speaker module over it is not:
            # no experiment of the reference sets this flag (config/network/wav2vec2_fc.yaml:12 `false`)
            raise NotImplementedError("wav2vec_feature_encoder_only=True: use models.wav2vec2.Wav2vecLiteWrapperModule "
                                      "directly; the speaker module over the CNN-only encoder is not built")
        # ``final_channel_mask_prob`` is accepted and has NO effect, like in the reference: its EmbeddingMasker gates
        # the channel mask on ``timestep_mask_prob``, which Wav2vec2FCModule hard-wires to 0 (quirk Q3,
        # ref: src/layers/embedding_masking.py:79, wav2vec2_fc.py:162-169)
        loss_fn = loss_fn_constructor()
        if isinstance(loss_fn, AngularAdditiveMarginSoftMaxLoss):
            loss, margin, scale = "aam", float(loss_fn.margin), float(loss_fn.scale)
        elif isinstance(loss_fn, CrossEntropyLoss):
            loss, margin, scale = "ce", 0.2, 30.0
        else:
            raise NotImplementedError(f"loss {type(loss_fn).__name__}: only AngularAdditiveMarginSoftMaxLoss and "
                                      "CrossEntropyLoss are on the hot path (w2v2_speaker_amd.optim.loss)")
        del loss_fn            # its sizes are placeholders (ref: config/optim/loss/aam_softmax.yaml), see module docstring
        self.model_cfg = W2V2Config.from_huggingface_id(cfg.wav2vec_hunggingface_id)
        self.num_speakers = num_speakers
        self.reg = Wav2Vec2RegularisationConfig(
            activation_dropout=cfg.activation_dropout, attention_dropout=cfg.attention_dropout,
            feat_proj_dropout=cfg.feat_proj_dropout, hidden_dropout=cfg.hidden_dropout, layerdrop=cfg.layerdrop,
            mask_feature_length=cfg.mask_feature_length, mask_feature_prob=cfg.mask_feature_prob,
            mask_time_length=cfg.mask_time_length, mask_time_prob=cfg.mask_time_prob)
        H = self.model_cfg.hidden_size
        self.stat_pool_dimension = (cfg.explicit_stat_pool_embedding_size
                                    if cfg.explicit_stat_pool_embedding_size is not None
                                    else H * _pool_width(cfg.stat_pooling_type))
        hidden = tuple(cfg.hidden_fc_layers_out)
        n_out = cfg.explicit_num_speakers if cfg.explicit_num_speakers else num_speakers
        # ref: :277-288 _determine_embedding_size
        if cfg.embedding_layer_idx < 0:
            self.embedding_size = self.stat_pool_dimension
        elif cfg.embedding_layer_idx < len(hidden):
            self.embedding_size = hidden[cfg.embedding_layer_idx]
        elif cfg.embedding_layer_idx == len(hidden) and loss == "ce":
            self.embedding_size = num_speakers
        else:
            raise ValueError("could not determine size of speaker embeddings")
        self.store = ParamStore(self.model_cfg, device, act_dtype, head=loss, num_speakers=n_out,
                                embed_dim=self.stat_pool_dimension, hidden_fc=hidden,
                                freeze_cnn=cfg.completely_freeze_feature_extractor,
                                attentive_pool="attentive" in (cfg.stat_pooling_type, cfg.test_stat_pooling_type))
        self.store.init_weights(init_seed)
        if pretrained_state_dict is not None:
            sd = (torch.load(pretrained_state_dict, map_location="cpu", weights_only=True)      # plain tensors only
                  if isinstance(pretrained_state_dict, str) else pretrained_state_dict)
            self.store.load_state_dict({k: v for k, v in sd.items()}, strict=False, prefix_model=True)
        elif not cfg.reset_weights:
            warnings.warn("Wav2vec2FCModule: reset_weights=False asks for the pretrained "
                          f"{cfg.wav2vec_hunggingface_id!r} weights, but none were given (pass pretrained_state_dict=; "
                          "there is no network access for from_pretrained): the model starts from a RANDOM "
                          "initialisation", stacklevel=2)
        self.loss, self.margin, self.scale = loss, margin, scale
        self.validation_pairs, self.test_pairs = validation_pairs or [], test_pairs or []
        self.evaluator = evaluator or CosineDistanceEvaluator(False, False, 0)
        self.schedule = OneCycle(max_lr=max_lr, total_steps=max_steps)
        self.process_group = process_group
        self._plans: "OrderedDict[Tuple, Plan]" = OrderedDict()
        self._trainers: Dict[Tuple, SpeakerTrainer] = {}
        self.steps = 0              # ref: counts backward calls since on_train_start (the freeze schedule)
        self.schedule_step = 0      # position in the learning-rate schedule (restored from a checkpoint)
        self._is_wav2vec_frozen = False
        self.device = torch.device(device)
        self.wav2vec = _Wav2vecHandle(self)
        self.test_with_ensemble = cfg.use_transformers_as_ensembles
        self.train_acc: Optional[torch.Tensor] = None

    @classmethod
    def from_config(cls, cfg: Wav2vec2FCModuleConfig, num_speakers: int, loss: str = "aam", aam_margin: float = 0.2,
                    aam_scale: float = 30.0, **kw) -> "Wav2vec2FCModule":
        """Short form for scripts and tests: the loss by name instead of a constructor."""
        assert loss in ("aam", "ce")
        dev = kw.get("device", "cuda")

        def ctor():
            if loss == "aam":      # placeholder sizes, like config/optim/loss/aam_softmax.yaml
                return AngularAdditiveMarginSoftMaxLoss(2, 2, margin=aam_margin, scale=aam_scale, device=dev,
                                                        act_dtype=torch.float32)
            return CrossEntropyLoss()
        return cls(None, cfg, num_speakers, ctor, kw.pop("validation_pairs", None), kw.pop("test_pairs", None),
                   kw.pop("evaluator", None), **kw)

    # ------------------------------------------------------------------ nn.Module surface over the flat arena
    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        if getattr(self, "_param_views", None) is None:
            self._param_views = {}
            for n in self.store.reference_parameter_order():
                p = torch.nn.Parameter(self.store.p(n), requires_grad=False)        # a view: shares the arena's memory
                if self.store.is_trainable(n):
                    p.grad = self.store.g(n)
                self._param_views[n] = p
        for n, p in self._param_views.items():
            yield (prefix + "." if prefix else "") + n, p       # same keys as state_dict() (save_checkpoint maps to the torch-1.9 names)

    def parameters(self, recurse: bool = True):
        for _, p in self.named_parameters(recurse=recurse):
            yield p

    def _apply(self, fn, recurse: bool = True):
        return self            # .to() / .cuda() / .half() / .float(): the engine keeps its device and precision

    def _set_body_trainable(self, flag: bool) -> None:
        """``wav2vec.model.{feature_projection,encoder}.requires_grad_``: only all-or-nothing freezes exist on this
        path (the reference's ``wav2vec_initially_frozen``)."""
        self._is_wav2vec_frozen = not flag

    def on_train_start(self) -> None:
        # ref: wav2vec2_fc.py:339-347
        self.steps = 0
        if self.cfg.wav2vec_initially_frozen:
            self.wav2vec.freeze()
        if self.cfg.completely_freeze_feature_extractor:
            self.wav2vec.model.feature_extractor.requires_grad_(False)

    def on_after_backward(self) -> None:
        # ref: wav2vec2_fc.py:349-361 -- num_frozen_steps counts backward calls
        self.steps += 1
        self.schedule_step += 1
        if (self._is_wav2vec_frozen and self.cfg.num_frozen_steps is not None
                and self.steps >= self.cfg.num_frozen_steps):
            self.wav2vec.unfreeze()
            if self.cfg.completely_freeze_feature_extractor:
                self.wav2vec.model.feature_extractor.requires_grad_(False)

    # ------------------------------------------------------------------ plans (bounded LRU cache)
    def _cached_plan(self, key: Tuple, build: Callable[[], Plan]) -> Plan:
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        plan = build()
        self._plans[key] = plan
        while len(self._plans) > MAX_PLANS:
            old, _ = self._plans.popitem(last=False)
            self._trainers.pop(old, None)
        return plan

    def _plan(self, batch: int, n: int, train: bool) -> Plan:
        pooling = self.cfg.stat_pooling_type if train else self.cfg.test_stat_pooling_type
        return self._cached_plan((batch, n, train, pooling), lambda: Plan(
            self.store, batch, n, train=train, reg=self.reg, pooling=pooling,
            insert_cls_token=(pooling == "first+cls"), aam_margin=self.margin, aam_scale=self.scale))

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
        """ref: :414-431 -- wav2vec2 -> statistics pooling -> (identity masker) -> hidden FC layers up to
        ``embedding_layer_idx``."""
        x = self._prep_input(input_tensor).to(self.device, torch.float32)
        plan = self._plan(x.shape[0], x.shape[1], False)
        plan.embed(x)
        emb = plan.speaker_embedding(self.cfg.embedding_layer_idx).clone()
        if plan.no_pool:               # NoPooling: [B, T, features], like the reference (its fc layers act on the last dim)
            emb = emb.view(plan.B, plan.T, -1)
        return emb

    def _linear(self, x: torch.Tensor, i: int, relu: bool) -> torch.Tensor:
        W, b = self.store.p(f"fc_list.{i}.0.weight"), self.store.p(f"fc_list.{i}.0.bias")
        out = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=self.device)
        ops.skinny_linear_fwd(x.float().contiguous(), W, b, out, ops.ACT_RELU if relu else ops.ACT_NONE)
        return out

    def compute_speaker_prediction(self, embedding_tensor: torch.Tensor) -> torch.Tensor:
        """ref: :400-412,433-438 -- the FC layers after ``embedding_layer_idx``; with AAM the classifier weight
        belongs to the loss, so the "prediction" of the default configuration is the embedding itself (quirk Q6)."""
        x = embedding_tensor.to(self.device)
        if x.dim() == 1:
            x = x[None]
        lead = None
        if x.dim() == 3:               # NoPooling: nn.Linear acts on the last dimension
            lead = x.shape[:2]
            x = x.reshape(-1, x.shape[-1])
        nh = len(self.cfg.hidden_fc_layers_out)
        for i in range(self.cfg.embedding_layer_idx + 1, nh):
            x = self._linear(x, i, relu=True)
        if self.loss == "ce" and self.cfg.embedding_layer_idx < nh:
            x = self._linear(x, nh, relu=False)
        if lead is not None:
            x = x.view(*lead, -1)
        return x.squeeze()

    def forward(self, input_tensor: torch.Tensor):
        embedding = self.compute_speaker_embedding(input_tensor)
        return embedding, self.compute_speaker_prediction(embedding)

    def generate_example_input(self, include_batch_dimension: bool, batch_size: Optional[int] = None):
        # ref: wav2vec2_fc.py:321-337
        shape = [batch_size, 16000] if include_batch_dimension else [16000]
        return torch.rand(size=shape)

    # ------------------------------------------------------------------ steps
    def broadcast_state(self, root: int = 0) -> None:
        """What PL's DDP does at ``trainer.fit`` plus the host half of a resume: every rank takes ``root``'s
        parameters, Adam moments, loss scale, Adam step counts, schedule position and freeze counter -- so a checkpoint
        loaded on rank 0 only leaves the replicas identical (under PL every rank restores the checkpoint itself)."""
        from ...trainer import BucketAllReducer
        red = BucketAllReducer(self.store, self.process_group)
        got = red.broadcast_parameters(root, [self.schedule_step, self.steps, int(self._is_wav2vec_frozen)])
        self.schedule_step, self.steps = got[0], got[1]
        self._is_wav2vec_frozen = bool(got[2])

    def training_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0,
                      optimizer_idx: Optional[int] = None):
        """forward + backward (+ all-reduce) + fused Adam; returns {"loss", "prediction", "train_acc"} (device
        tensors; train_acc = fraction of the batch whose arg-max prediction is the label, the quantity the
        reference feeds torchmetrics.Accuracy, ref: speaker_recognition_module.py:296-307)."""
        x = self._prep_input(batch.network_input).to(self.device, torch.float32)
        label = batch.ground_truth.to(self.device)
        pkey = (x.shape[0], x.shape[1], True, self.cfg.stat_pooling_type)
        plan = self._plan(x.shape[0], x.shape[1], True)
        if pkey not in self._trainers:
            self._trainers[pkey] = SpeakerTrainer(self.store, plan, self.schedule, process_group=self.process_group)
        tr = self._trainers[pkey]
        tr.step = self.schedule_step
        if self._is_wav2vec_frozen:
            # frozen network = eval-mode forward (PL freeze()), head-only backward + Adam
            noreg = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                                                 hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
            fplan = self._cached_plan((x.shape[0], x.shape[1], "frozen"), lambda: Plan(
                self.store, x.shape[0], x.shape[1], train=True, reg=noreg, pooling=self.cfg.stat_pooling_type,
                insert_cls_token=(self.cfg.stat_pooling_type == "first+cls"), aam_margin=self.margin,
                aam_scale=self.scale))
            loss, pred = tr.train_step_frozen_encoder(fplan, x, label)
            head = fplan.head
        else:
            loss, pred = tr.train_step(x, label)
            head = plan.head
        self.train_acc = head.correct.mean()
        self.on_after_backward()
        return {"loss": loss, "prediction": pred, "train_acc": self.train_acc}

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

    def compute_ensemble_embedding(self, input_tensor: torch.Tensor):
        """ref: wav2vec2_fc.py:440-463 -- list of ``num_ensembles`` pooled embeddings, one per hidden state of the
        last transformer layers (``use_transformers_as_ensembles``); scored by CosineDistanceEvaluator as the mean
        of the per-layer cosine scores."""
        x = self._prep_input(input_tensor)
        plan = self._cached_plan(("ensemble", x.shape[0], x.shape[1]), lambda: Plan(
            self.store, x.shape[0], x.shape[1], train=False, reg=self.reg, pooling=self.cfg.stat_pooling_type,
            keep_hidden_states=True, insert_cls_token=(self.cfg.stat_pooling_type == "first+cls")))
        return plan.ensemble_embeddings(x.to(self.device), self.cfg.num_ensembles)

    # ------------------------------------------------------------------ checkpoints (reference key names)
    def state_dict(self, *args, **kwargs):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        self.store.load_state_dict(sd, strict=strict, prefix_model=False)

    def save_checkpoint(self, path: str, legacy_weight_norm_names: bool = True) -> None:
        """A PL-1.4-style checkpoint file (the reference pins pytorch-lightning 1.4.5): a pickled dict whose
        ``state_dict`` uses the reference's parameter and buffer names (``wav2vec.model.<HF name>``,
        ``loss_fn.fc_weights`` / ``fc_list.{i}.0.*``, ``stat_pooling.pooling_layer.*`` incl. the BatchNorm running
        statistics).  ``legacy_weight_norm_names`` (default) writes the pos-conv weight-norm pair as
        ``...conv.weight_g`` / ``...conv.weight_v`` -- the names of the reference's own stack (torch 1.9 /
        transformers ^4.8); its ``load_from_checkpoint(strict=False)`` would silently DROP the torch >= 2.1 names
        ``parametrizations.weight.original0/1`` (load accepts both).  ``optimizer_states[0]`` is a
        ``torch.optim.Adam.state_dict()`` in the reference's parameter order (ParamStore.torch_adam_state) and
        ``lr_schedulers[0]`` the ``OneCycleLR`` fields a resume needs; the engine's own extras (fp16 loss-scale
        record, freeze-schedule counters) travel under ``w2v2_amd``.  What is NOT claimed: torchmetrics / callback
        states of a PL ``Trainer`` -- a ``Trainer`` resume restores weights, optimiser moments and schedule position."""
        sd = self.state_dict()
        if legacy_weight_norm_names:
            sd = OrderedDict((ParamStore.legacy_key(k), v) for k, v in sd.items())
        # torch's groups hold the values of the NEXT optimiser step (OneCycleLR.step() writes them right after a step)
        sch = self.schedule
        lr, beta1 = sch.at(min(self.schedule_step, sch.total_steps - 1))
        z = lambda t: None if t is None else t.detach().clone().cpu()
        initial_lr = sch.max_lr / sch.div_factor
        # torch.optim.lr_scheduler.OneCycleLR.state_dict() (every key it reads back in load_state_dict / step)
        sched_state = {"total_steps": sch.total_steps,
                       "_schedule_phases": [
                           {"end_step": float(sch.pct_start * sch.total_steps) - 1, "start_lr": "initial_lr", "end_lr": "max_lr",
                            "start_momentum": "max_momentum", "end_momentum": "base_momentum"},
                           {"end_step": sch.total_steps - 1, "start_lr": "max_lr", "end_lr": "min_lr",
                            "start_momentum": "base_momentum", "end_momentum": "max_momentum"}],
                       "_anneal_func_type": "cos", "cycle_momentum": True, "use_beta1": True, "base_lrs": [initial_lr],
                       "last_epoch": self.schedule_step, "_step_count": self.schedule_step + 1, "_is_initial": False,
                       "_get_lr_called_within_step": False, "_last_lr": [lr]}
        torch.save({"state_dict": sd, "global_step": self.schedule_step, "epoch": 0,
                    "pytorch-lightning_version": "1.4.5",
                    "optimizer_states": [self.store.torch_adam_state(lr, (beta1, 0.999), 1e-8, one_cycle=sch)],
                    "lr_schedulers": [sched_state],
                    "w2v2_amd": {"loss_scaler": z(self.store.scaler), "steps": self.steps,
                                 "is_wav2vec_frozen": self._is_wav2vec_frozen},
                    "hyper_parameters": {"num_speakers": self.num_speakers, "loss": self.loss}}, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, strict: bool = False, **kwargs) -> "Wav2vec2FCModule":
        """ref: src/main.py:272-283 (``network_class.load_from_checkpoint(path, strict=False, **kwargs)``): construct
        from the reference's kwargs, then load ``checkpoint["state_dict"]``.  Non-strict: unknown keys are skipped,
        tensors whose shape differs from this module's (a head re-sized through ``explicit_num_speakers``) and
        missing ones keep their fresh initialisation.  Optimiser moments, the loss scale and the schedule position
        resume when the checkpoint has them and the arena still has the same size."""
        kwargs.setdefault("hyperparameters_to_save", None)
        kwargs_trusted = bool(kwargs.pop("trust_checkpoint_pickle", False))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # weights come from the checkpoint, not from "pretrained"
            module = cls(**kwargs)
        ckpt = _load_checkpoint_file(checkpoint_path, kwargs_trusted)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        if not strict:
            shapes = module.store.shapes
            canon = lambda k: next((k[: -len(o)] + n for o, n in _WN_LEGACY.items() if k.endswith(o)), k)
            sd = {k: v for k, v in sd.items() if canon(k) not in shapes or tuple(v.shape) == tuple(shapes[canon(k)])}
        module.load_state_dict(sd, strict=strict)
        if isinstance(ckpt, dict):
            module.schedule_step = int(ckpt.get("global_step", 0))
            fs = ckpt.get("w2v2_amd") or ckpt.get("freeze_schedule") or {}
            module.steps = int(fs.get("steps", module.schedule_step))
            module._is_wav2vec_frozen = bool(fs.get("is_wav2vec_frozen", False))
            if module.store.scaler is not None and fs.get("loss_scaler") is not None:
                rec = torch.as_tensor(fs["loss_scaler"]).to(module.store.device, torch.float32).reshape(-1)
                module.store.scaler[:min(rec.numel(), module.store.scaler.numel())] = rec[:module.store.scaler.numel()]
            for ost in ckpt.get("optimizer_states") or []:
                try:
                    if "param_groups" in ost:             # torch.optim.Adam.state_dict() (reference parameter order)
                        module.store.load_torch_adam_state(ost)
                    else:                                 # round-2 files: flat moment arenas
                        module.store.load_optimizer_state(ost)
                except (ValueError, IndexError):
                    if strict:
                        raise
        return module
