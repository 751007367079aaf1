"""Host-side SpecAugment time-mask sampler (HF ``_compute_mask_indices``, HF:101-217), restated.

Same numpy global-RNG call sequence as HF (one ``np.random.rand(1)`` for the probabilistic rounding,
then one ``np.random.choice(..., replace=False)`` per utterance), so ``np.random.seed(s)`` reproduces
HF's masks bit-for-bit (pinned by tests/golden/g8_optim.npz).  No attention_mask on this path."""
from __future__ import annotations

import numpy as np


def compute_mask_indices(shape, mask_prob: float, mask_length: int, min_masks: int = 0,
                         rng=np.random) -> np.ndarray:
    batch_size, sequence_length = shape
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > sequence_length:
        raise ValueError(f"`mask_length` has to be smaller than `sequence_length`, but got `mask_length`: "
                         f"{mask_length} and `sequence_length`: {sequence_length}`")
    epsilon = rng.rand(1).item()
    n = int(mask_prob * sequence_length / mask_length + epsilon)
    n = max(n, min_masks)
    if n * mask_length > sequence_length:
        n = sequence_length // mask_length
    if sequence_length - (mask_length - 1) < n:
        n = max(sequence_length - (mask_length - 1), 0)
    mask = np.zeros((batch_size, sequence_length), dtype=bool)
    if n == 0:
        return mask
    starts = np.stack([rng.choice(np.arange(sequence_length - (mask_length - 1)), n, replace=False)
                       for _ in range(batch_size)])
    idx = (starts[:, :, None] + np.arange(mask_length)[None, None, :]).reshape(batch_size, -1)
    idx = np.minimum(idx, sequence_length - 1)
    np.put_along_axis(mask, idx, True, -1)
    return mask
