"""Mirror of ref: src/evaluation/speaker/{speaker_recognition_evaluator,cosine_distance}.py: trial scoring by
cosine similarity, ``(s+1)/2`` clipped to [0,1] (evaluator.py:81), then EER / minDCF."""
from __future__ import annotations

import random
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


def compute_mean_std_batch(all_tensors: torch.Tensor):
    """ref: speaker_recognition_evaluator.py:154-159 -- per-dimension mean and UNBIASED std over [NUM_SAMPLES, EMBEDDING_SIZE]."""
    std, mean = torch.std_mean(all_tensors, dim=0)
    return mean, std


def center_batch(embedding_tensor: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """ref: speaker_recognition_evaluator.py:162-167 -- despite the name this is a per-dimension z-score:
    the reference divides by ``std + 1e-12`` as well as subtracting the mean."""
    return (embedding_tensor - mean) / (std + 1e-12)


def length_norm_batch(embedding_tensor: torch.Tensor) -> torch.Tensor:
    """ref: speaker_recognition_evaluator.py:170-172."""
    return torch.nn.functional.normalize(embedding_tensor, dim=1)


def compute_cosine_scores(left_samples: torch.Tensor, right_samples: torch.Tensor) -> List[float]:
    """ref: cosine_distance.py:235-241 (``torch.nn.CosineSimilarity()``: dim 1, eps 1e-8)."""
    return torch.nn.functional.cosine_similarity(left_samples, right_samples, dim=1, eps=1e-8) \
        .detach().cpu().numpy().tolist()


def compute_non_pooled_cosine_scores(left_tensor: torch.Tensor, right_tensor: torch.Tensor) -> float:
    """ref: cosine_distance.py:203-232 -- a sample whose embedding is 2-D ([frames, features], the ``NoPooling`` head):
    at most 50 frames per side, drawn with the GLOBAL ``random`` module (left side first, so a caller that seeds
    ``random`` sees the reference's draws), and the score is the mean of all pairwise frame cosines.  Element (i, j) of
    the broadcast below is element ``i * p2 + j`` of the reference's repeat_interleave / repeat pair, so the mean runs
    over the same values in the same order."""
    p1, p2 = left_tensor.shape[0], right_tensor.shape[0]
    left = left_tensor[random.sample(range(p1), min(50, p1)), :]
    right = right_tensor[random.sample(range(p2), min(50, p2)), :]
    with torch.no_grad():
        n1, n2, d = left.shape[0], right.shape[0], left.shape[1]
        score = torch.nn.functional.cosine_similarity(
            left[:, None, :].expand(n1, n2, d).reshape(n1 * n2, d),
            right[None, :, :].expand(n1, n2, d).reshape(n1 * n2, d), dim=1, eps=1e-8)
    return torch.mean(score).detach().cpu().numpy().tolist()


class CosineDistanceEvaluator:
    """ref: cosine_distance.py:66-201.  Defaults = config/evaluator/cosine_distance.yaml (no centering, no length
    norm, max_num_training_samples 0); the non-default branches follow the reference as well: centering is
    ``(x - mean) / (std + 1e-12)`` with the statistics of ``fit_parameters``, then optional length norm, then the
    cosine; 2-D embeddings take the non-pooled scoring (no centering / length norm there, as in the reference)."""

    def __init__(self, center_before_scoring: bool = False, length_norm_before_scoring: bool = False,
                 max_num_training_samples: int = 0):
        self.center_before_scoring = center_before_scoring
        self.length_norm_before_scoring = length_norm_before_scoring
        self.max_num_training_samples = max_num_training_samples
        self.mean = None
        self.std = None

    def _using_parameters(self) -> bool:
        return self.center_before_scoring

    def fit_parameters(self, embedding_tensors: List[torch.Tensor], _label_tensors=None):
        if not self._using_parameters():
            return
        if len(embedding_tensors) <= 2:
            raise ValueError("mean/std calculation requires more than 2 samples")
        self.mean, self.std = compute_mean_std_batch(torch.stack(embedding_tensors, dim=0))

    def reset_parameters(self):
        """ref: cosine_distance.py:99-104."""
        if not self._using_parameters():
            return
        self.mean = None
        self.std = None

    def _transform_pairs_to_tensor(self, pairs: List[Tuple[EmbeddingSample, EmbeddingSample]]):
        """ref: speaker_recognition_evaluator.py:123-137."""
        return (torch.stack([torch.as_tensor(a.embedding) for a, _ in pairs]),
                torch.stack([torch.as_tensor(b.embedding) for _, b in pairs]))

    def _compute_prediction_scores(self, pairs: List[Tuple[EmbeddingSample, EmbeddingSample]]) -> List[float]:
        first = pairs[0][0].embedding
        if isinstance(first, list):                     # ref: cosine_distance.py:110-112 (ensemble of layers)
            return self._compute_ensemble_prediction_scores(pairs)
        if len(first.shape) == 2:                       # ref: cosine_distance.py:114-116 (non-pooled embeddings)
            return self._compute_non_pooled_prediction_scored(pairs)
        left, right = self._transform_pairs_to_tensor(pairs)
        if self.center_before_scoring:
            if self.mean is None or self.std is None:   # the reference dies on ``tensor - None`` here; say why
                raise TypeError("center_before_scoring=True but fit_parameters() has not been called "
                                "(mean / std are None)")
            left = center_batch(left, self.mean, self.std)
            right = center_batch(right, self.mean, self.std)
        if self.length_norm_before_scoring:
            left = length_norm_batch(left)
            right = length_norm_batch(right)
        return compute_cosine_scores(left, right)

    def _compute_non_pooled_prediction_scored(self, pairs: List[Tuple[EmbeddingSample, EmbeddingSample]]) -> List[float]:
        """ref: cosine_distance.py:187-200 (name kept, typo included: callers may reach for it)."""
        return [compute_non_pooled_cosine_scores(a.embedding, b.embedding) for a, b in pairs]

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
