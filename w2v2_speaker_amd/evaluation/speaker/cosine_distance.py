"""Mirror of ref: src/evaluation/speaker/{speaker_recognition_evaluator,cosine_distance}.py: trial scoring by
cosine similarity, ``(s+1)/2`` clipped to [0,1] (evaluator.py:81), then EER / minDCF."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple, Union
from warnings import warn

import numpy as np
import torch

from ...eval_metrics import calculate_eer, calculate_mdc


@dataclass
class EvaluationPair:
    same_speaker: bool
    sample1_id: str
    sample2_id: str


@dataclass
class EmbeddingSample:
    sample_id: str
    embedding: Union[torch.Tensor, List[torch.Tensor]]


def compute_cosine_scores(left_samples: torch.Tensor, right_samples: torch.Tensor) -> List[float]:
    return torch.nn.functional.cosine_similarity(left_samples.float(), right_samples.float(), dim=1) \
        .detach().cpu().numpy().tolist()


class CosineDistanceEvaluator:
    """ref: cosine_distance.py:66-201 with the default configuration of the hot path
    (config/evaluator/cosine_distance.yaml: no centering, no length-norm, max_num_training_samples=0)."""

    def __init__(self, center_before_scoring: bool = False, length_norm_before_scoring: bool = False,
                 max_num_training_samples: int = 0):
        self.center_before_scoring = center_before_scoring
        self.length_norm_before_scoring = length_norm_before_scoring
        self.max_num_training_samples = max_num_training_samples
        self.mean = None
        self.std = None

    def fit_parameters(self, embedding_tensors: List[torch.Tensor], _label_tensors=None):
        if not self.center_before_scoring:
            return
        if len(embedding_tensors) <= 2:
            raise ValueError("mean/std calculation requires more than 2 samples")
        allt = torch.stack(embedding_tensors, dim=0)
        self.std, self.mean = torch.std_mean(allt, dim=0)

    def _prep(self, t: torch.Tensor) -> torch.Tensor:
        if self.center_before_scoring and self.mean is not None:
            t = t - self.mean
        if self.length_norm_before_scoring:
            t = torch.nn.functional.normalize(t, dim=1)
        return t

    def _compute_prediction_scores(self, pairs: List[Tuple[EmbeddingSample, EmbeddingSample]]) -> List[float]:
        if isinstance(pairs[0][0].embedding, list):     # ref: cosine_distance.py:110-112 (ensemble of layers)
            return self._compute_ensemble_prediction_scores(pairs)
        left = self._prep(torch.stack([torch.as_tensor(a.embedding).flatten().float().cpu() for a, _ in pairs]))
        right = self._prep(torch.stack([torch.as_tensor(b.embedding).flatten().float().cpu() for _, b in pairs]))
        return compute_cosine_scores(left, right)

    def _compute_ensemble_prediction_scores(self, pairs: List[Tuple[EmbeddingSample, EmbeddingSample]]) -> List[float]:
        """ref: cosine_distance.py:134-185 -- every sample carries a LIST of embeddings (one per hidden state,
        Wav2vec2FCModule.compute_ensemble_embedding); the score of a pair is the mean of the per-member scores."""
        n = len(pairs[0][0].embedding)
        for s1, s2 in pairs:
            if not isinstance(s1.embedding, list) or not isinstance(s2.embedding, list):
                raise ValueError("not every embedding sample is an ensemble")
            if len(s1.embedding) != n or len(s2.embedding) != n:
                raise ValueError(f"expected each list to have len num_ensembles={n}")
        member_scores = [self._compute_prediction_scores(
            [(EmbeddingSample(sample_id=a.sample_id, embedding=a.embedding[i]),
              EmbeddingSample(sample_id=b.sample_id, embedding=b.embedding[i])) for a, b in pairs]) for i in range(n)]
        combined = []
        for idx in range(len(pairs)):
            score = 0
            for i in range(n):                       # same accumulation order as the reference
                score += member_scores[i][idx] * (1 / n)
            combined.append(score)
        return combined

    def evaluate(self, pairs: List[EvaluationPair], samples: List[EmbeddingSample]):
        sample_map = {}
        for sample in samples:
            if sample.sample_id in sample_map:
                raise ValueError(f"duplicate key {sample.sample_id}")
            sample_map[sample.sample_id] = sample
        gt, pp = [], []
        for pair in pairs:
            if pair.sample1_id not in sample_map or pair.sample2_id not in sample_map:
                warn(f"{pair.sample1_id} or {pair.sample2_id} not in sample_map")
                return {"eer": -1, "eer_threshold": -1, "mdc": -1, "mdc_threshold": -1}
            gt.append(1 if pair.same_speaker else 0)
            pp.append((sample_map[pair.sample1_id], sample_map[pair.sample2_id]))
        scores = np.clip((np.array(self._compute_prediction_scores(pp)) + 1) / 2, 0, 1).tolist()
        try:
            eer, eer_threshold = calculate_eer(gt, scores, pos_label=1)
        except (ValueError, ZeroDivisionError) as e:       # reference falls back to a very bad score
            print(f"EER calculation had {e}")
            eer, eer_threshold = 1, 1337
        try:
            mdc, mdc_threshold = calculate_mdc(gt, scores)
        except (ValueError, ZeroDivisionError) as e:
            print(f"mdc calculation had {e}")
            mdc, mdc_threshold = 1, 1337
        return {"eer": eer, "eer_threshold": eer_threshold, "mdc": mdc, "mdc_threshold": mdc_threshold}
