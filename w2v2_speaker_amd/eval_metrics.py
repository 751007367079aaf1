"""Mirror of ref: src/eval_metrics.py -- EER (ROC + brentq root of 1 - x - tpr(x), :54-79) and minDCF
(:90-206, Kaldi-style sweep).  Host numpy/scipy (seconds per evaluation)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
from scipy.interpolate import interp1d
from scipy.optimize import brentq


def _verify_correct_scores(groundtruth_scores, predicted_scores):
    if len(groundtruth_scores) != len(predicted_scores):
        raise ValueError(f"length of input lists should match, while groundtruth_scores={len(groundtruth_scores)} "
                         f"and predicted_scores={len(predicted_scores)}")
    if not all(np.isin(groundtruth_scores, [0, 1])):
        raise ValueError(f"groundtruth values should be either 0 and 1, while they are actually one of "
                         f"{np.unique(groundtruth_scores)}")


def roc_curve(y_true, y_score, pos_label=1):
    """sklearn.metrics.roc_curve (drop_intermediate=True) semantics, which the reference calls."""
    y_true = (np.asarray(y_true) == pos_label)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    idx = np.r_[np.where(np.diff(y_score))[0], y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    thr = y_score[idx]
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps, fps, thr = np.r_[0, tps], np.r_[0, fps], np.r_[np.inf, thr]
    if fps[-1] <= 0 or tps[-1] <= 0:
        raise ValueError("need both positive and negative trials")
    return fps / fps[-1], tps / tps[-1], thr


def calculate_eer(groundtruth_scores: List[int], predicted_scores: List[float], pos_label: int = 1):
    _verify_correct_scores(groundtruth_scores, predicted_scores)
    if not all(np.isin([pos_label], [0, 1])):
        raise ValueError(f"The positive label should be either 0 or 1, not {pos_label}")
    fpr, tpr, thresholds = roc_curve(groundtruth_scores, predicted_scores, pos_label=pos_label)
    eer = brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0)
    thresh = interp1d(fpr, thresholds)(eer).item()
    return eer, thresh


def calculate_mdc(groundtruth_scores: List[int], predicted_scores: List[float], c_miss: float = 1,
                  c_fa: float = 1, p_target: float = 0.05) -> Tuple[float, float]:
    _verify_correct_scores(groundtruth_scores, predicted_scores)
    if c_miss < 1:
        raise ValueError(f"c_miss={c_miss} should be >= 1")
    if c_fa < 1:
        raise ValueError(f"c_fa={c_fa} should be >= 1")
    if p_target < 0 or p_target > 1:
        raise ValueError(f"p_target={p_target} should be between 0 and 1")
    gt = np.asarray(groundtruth_scores, dtype=np.float64)
    sc = np.asarray(predicted_scores, dtype=np.float64)
    order = np.argsort(sc, kind="stable")
    gt, thr = gt[order], sc[order]
    fnrs = np.cumsum(gt) / gt.sum()
    fprs = 1.0 - np.cumsum(1.0 - gt) / (len(gt) - gt.sum())
    c_det = c_miss * fnrs * p_target + c_fa * fprs * (1 - p_target)
    i = int(np.argmin(c_det))
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return float(c_det[i] / c_def), float(thr[i])
