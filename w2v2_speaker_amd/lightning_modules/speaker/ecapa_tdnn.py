"""Mirror of ``src/lightning_modules/speaker/ecapa_tdnn.py`` (EcapaTDNNModuleConfig :26-48, EcapaTdnnModule :51-137)
on the HIP path (w2v2_speaker_amd/ecapa.py).  Same config field names, same method names and argument meaning."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ...ecapa import EcapaConfig, EcapaPlan, EcapaStore, EcapaTrainer
from ...optim.schedule import OneCycle
from .wav2vec2_fc import SpeakerClassificationDataBatch


@dataclass
class EcapaTDNNModuleConfig:
    """ref: ecapa_tdnn.py:26-48 / config/network/ecapa_tdnn.yaml."""
    input_mel_coefficients: int = 40
    lin_neurons: int = 192
    channels: List[int] = field(default_factory=lambda: [1024, 1024, 1024, 1024, 3072])
    kernel_sizes: List[int] = field(default_factory=lambda: [5, 3, 3, 3, 1])
    dilations: List[int] = field(default_factory=lambda: [1, 2, 3, 4, 1])
    attention_channels: int = 128
    res2net_scale: int = 8
    se_channels: int = 128
    global_context: bool = True
    pretrained_weights_path: Optional[str] = None
    explicit_stat_pool_embedding_size: Optional[int] = None
    explicit_num_speakers: Optional[int] = None


class EcapaTdnnModule:
    def __init__(self, hyperparameters_to_save, cfg: EcapaTDNNModuleConfig, num_speakers: int,
                 loss_fn_constructor: Callable[[], object], validation_pairs=None, test_pairs=None, evaluator=None, *,
                 device="cuda", act_dtype: torch.dtype = torch.bfloat16, max_lr: float = 1e-3,
                 max_steps: int = 100_000, init_seed: int = 20211):
        """Positional arguments = ref: ecapa_tdnn.py:51-62 (what src/main.py:256-285 passes to every network class).
        ``loss_fn_constructor`` is called once and read for its type and hyper-parameters: the engine runs the ECAPA
        model under AAM-softmax (``skip_classifier`` of ref :93-95; the paper's configuration,
        config/experiment/speaker_ecapa_tdnn.yaml) -- the cross-entropy + cosine ``Classifier`` variant is not on the
        path.  ``act_dtype=torch.float32`` is the reference's own precision for this model (``precision: 32``)."""
        from ...evaluation.speaker.cosine_distance import CosineDistanceEvaluator
        from ...optim.loss import AngularAdditiveMarginSoftMaxLoss
        loss_fn = loss_fn_constructor()
        if not isinstance(loss_fn, AngularAdditiveMarginSoftMaxLoss):
            raise NotImplementedError(f"loss {type(loss_fn).__name__}: the ECAPA path runs under "
                                      "AngularAdditiveMarginSoftMaxLoss (speechbrain's Classifier + CE is outside it)")
        aam_margin, aam_scale = float(loss_fn.margin), float(loss_fn.scale)
        del loss_fn
        self.hyperparameters_to_save = hyperparameters_to_save
        self.validation_pairs, self.test_pairs = validation_pairs or [], test_pairs or []
        self.evaluator = evaluator or CosineDistanceEvaluator(False, False, 0)
        if not cfg.global_context:
            raise NotImplementedError("global_context=False (the reference config sets True)")
        self.cfg = cfg
        self.embedding_size = cfg.lin_neurons
        self.num_speakers = cfg.explicit_num_speakers or num_speakers
        self.model_cfg = EcapaConfig(cfg.input_mel_coefficients, cfg.lin_neurons, tuple(cfg.channels),
                                     tuple(cfg.kernel_sizes), tuple(cfg.dilations), cfg.attention_channels,
                                     cfg.res2net_scale, cfg.se_channels)
        self.store = EcapaStore(self.model_cfg, device, act_dtype, num_speakers=self.num_speakers)
        self.store.init_weights(init_seed)
        if cfg.pretrained_weights_path is not None:       # ref :88-91: a bare ECAPA_TDNN state dict
            sd = torch.load(cfg.pretrained_weights_path, map_location="cpu", weights_only=True)
            self.store.load_state_dict(dict(sd), strict=False)    # incl. every BatchNorm running_mean / running_var
        self.margin, self.scale = aam_margin, aam_scale
        self.schedule = OneCycle(max_lr=max_lr, total_steps=max_steps)
        self.skip_classifier = True                        # AAM owns the classifier weight (ref :93-95, :129-131)
        self.device = torch.device(device)
        self._plans: Dict[Tuple, EcapaPlan] = {}
        self._trainers: Dict[Tuple, EcapaTrainer] = {}
        self.steps = 0

    @classmethod
    def from_config(cls, cfg: EcapaTDNNModuleConfig, num_speakers: int, aam_margin: float = 0.2,
                    aam_scale: float = 30.0, **kw) -> "EcapaTdnnModule":
        """Short form for scripts and tests: the AAM loss by its two hyper-parameters instead of a constructor."""
        from ...optim.loss import AngularAdditiveMarginSoftMaxLoss
        dev = kw.get("device", "cuda")
        ctor = lambda: AngularAdditiveMarginSoftMaxLoss(2, 2, margin=aam_margin, scale=aam_scale, device=dev,
                                                        act_dtype=torch.float32)
        return cls(None, cfg, num_speakers, ctor, kw.pop("validation_pairs", None), kw.pop("test_pairs", None),
                   kw.pop("evaluator", None), **kw)

    def _plan(self, batch: int, frames: int, train: bool) -> EcapaPlan:
        key = (batch, frames, train)
        if key not in self._plans:
            self._plans[key] = EcapaPlan(self.store, batch, frames, train=train, aam_margin=self.margin,
                                         aam_scale=self.scale)
        return self._plans[key]

    def generate_example_input(self, include_batch_dimension: bool, batch_size: Optional[int] = None):
        # ref :97-108: [BATCH_SIZE, NUMBER_OF_WINDOWS, NUMBER_OF_MEL_COEFFICIENTS]
        shape = [batch_size, 100, self.cfg.input_mel_coefficients] if include_batch_dimension else \
            [100, self.cfg.input_mel_coefficients]
        return torch.rand(size=shape)

    def compute_speaker_embedding(self, input_tensor: torch.Tensor) -> torch.Tensor:
        # ref :110-118
        x = input_tensor if input_tensor.dim() == 3 else input_tensor[None]
        x = x.to(self.device, torch.float32)
        return self._plan(x.shape[0], x.shape[1], False).embed(x).clone()

    def compute_speaker_prediction(self, embedding_tensor: torch.Tensor) -> torch.Tensor:
        return embedding_tensor.squeeze()                  # ref :120-122 under AAM

    def forward(self, input_tensor: torch.Tensor):
        embedding = self.compute_speaker_embedding(input_tensor)
        return embedding, self.compute_speaker_prediction(embedding)

    __call__ = forward

    def training_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0,
                      optimizer_idx: Optional[int] = None):
        x = batch.network_input.to(self.device, torch.float32)
        label = batch.ground_truth.to(self.device)
        key = (x.shape[0], x.shape[1])
        if key not in self._trainers:
            self._trainers[key] = EcapaTrainer(self.store, self._plan(x.shape[0], x.shape[1], True), self.schedule)
        tr = self._trainers[key]
        tr.step = self.steps
        loss, pred = tr.train_step(x, label)
        self.steps += 1
        return {"loss": loss, "prediction": pred}

    def validation_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0):
        return {"embedding": self.compute_speaker_embedding(batch.network_input).detach().to("cpu"),
                "sample_id": batch.keys}

    def test_step(self, batch: SpeakerClassificationDataBatch, batch_idx: int = 0):
        if batch.batch_size != 1:
            raise ValueError("expecting a batch size of 1 for evaluating speaker embeddings")
        return self.validation_step(batch, batch_idx)

    def _evaluate(self, outputs, pairs):
        from ...evaluation.speaker.cosine_distance import EmbeddingSample
        samples = [EmbeddingSample(sample_id=k, embedding=o["embedding"][i]) for o in outputs
                   for i, k in enumerate(o["sample_id"])]
        return self.evaluator.evaluate(pairs, samples)

    def validation_epoch_end(self, outputs):
        return self._evaluate(outputs, self.validation_pairs)

    def test_epoch_end(self, outputs):
        return self._evaluate(outputs, self.test_pairs)

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict(sd, strict=strict)
