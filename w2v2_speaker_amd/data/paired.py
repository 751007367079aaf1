"""Pair batcher of the paired-input (w2v2-bce) pipeline.

ref: src/data/modules/speaker/voxceleb.py:1065-1387 (``PairedBatchProcessor``), training_batch_speaker.py:125-215
(paired sample / batch + the stacking collate), voxceleb.py:1393-1405 (trial-list reader).

Two modes, as in the reference:
* ``generate`` (training): samples arrive speaker-grouped (``sequential_same_speaker_samples`` consecutive samples
  of one speaker); once the queue holds a batch worth of samples, ``batch_size / sequential`` speakers are drawn
  (without replacement, weight 2**(samples of that speaker in the queue)), ``sequential`` samples of each are taken
  at random, and ``round(ratio * batch_size)`` positive + the rest negative ORDERED pairs are formed from them, then
  shuffled.  The module-level ``random`` is used with the same sequence of calls as the reference, so a seeded run
  draws the same pairs.  Two reference quirks are kept because they change what a seed produces: a negative pair is
  only checked for duplication against the POSITIVE list (duplicate negatives can occur), and an empty sample list
  only counts a "fail" without skipping the draw.
* ``reproduce`` (validation / test): all samples are collected by key and the given trial list is replayed in order,
  in batches of ``batch_size`` (last batch ragged).
"""
from __future__ import annotations

import dataclasses
import pathlib
import random
from collections import defaultdict
from typing import Callable, Dict, Iterable, Iterator, List, NamedTuple, Optional

import torch

from ..lightning_modules.speaker.wav2vec2_paired_input import PairedSpeakerClassificationDataBatch
from .pipeline import SpeakerClassificationDataSample


class EvaluationPair(NamedTuple):
    """ref: src/evaluation/speaker/speaker_recognition_evaluator.py (EvaluationPair)."""
    same_speaker: bool
    sample1_id: str
    sample2_id: str


@dataclasses.dataclass
class PairedSpeakerClassificationDataSample:
    primary_key: str
    secondary_key: str
    primary_input: torch.Tensor
    secondary_input: torch.Tensor
    ground_truth: int
    side_info: Optional[object] = None


def paired_default_collate_fn(lst: List[PairedSpeakerClassificationDataSample]) -> PairedSpeakerClassificationDataBatch:
    """Stack equally long (cropped) inputs; ``squeeze`` as the reference does, so [1, N] waveforms become [B, N]."""
    return PairedSpeakerClassificationDataBatch(
        batch_size=len(lst),
        primary_keys=[s.primary_key for s in lst],
        primary_network_input=torch.stack([s.primary_input.squeeze() for s in lst]),
        secondary_keys=[s.secondary_key for s in lst],
        secondary_network_input=torch.stack([s.secondary_input.squeeze() for s in lst]),
        ground_truth=torch.tensor([int(s.ground_truth) for s in lst], dtype=torch.int64))


def read_test_pairs_file(path) -> Iterator[EvaluationPair]:
    """Lines ``<0|1> <utt a> <utt b>``; lines with fewer than two blanks are skipped (voxceleb.py:1393-1405)."""
    with pathlib.Path(path).open("r") as f:
        for line in f:
            line = line.strip()
            if line.count(" ") < 2:
                continue
            gt, a, b = line.split(" ")
            yield EvaluationPair(bool(int(gt)), a, b)


def _make_pair(a: SpeakerClassificationDataSample, b: SpeakerClassificationDataSample, same: int):
    return PairedSpeakerClassificationDataSample(primary_key=a.key, secondary_key=b.key, primary_input=a.network_input,
                                                 secondary_input=b.network_input, ground_truth=same)


class PairedBatchProcessor:
    def __init__(self, batch_size: int, max_queue_size: int, mode: str, sequential_same_speaker_samples: int,
                 collate_fn: Callable[[List[PairedSpeakerClassificationDataSample]],
                                      PairedSpeakerClassificationDataBatch] = paired_default_collate_fn,
                 pos_neg_training_batch_ratio: Optional[float] = None,
                 pairs: Optional[List[EvaluationPair]] = None, fixed_random_seed: bool = False,
                 yield_limit: Optional[int] = None):
        if mode not in ("generate", "reproduce"):
            raise ValueError(f"mode={mode!r} should be `generate` or `reproduce`")
        if batch_size > max_queue_size:
            raise ValueError(f"cannot generate batches of size {batch_size} with a max queue size of {max_queue_size}")
        if mode == "generate":
            if pos_neg_training_batch_ratio is None:
                raise ValueError("mode 'generate' needs pos_neg_training_batch_ratio")
            if batch_size % sequential_same_speaker_samples != 0:
                raise ValueError(f"batch_size={batch_size} must be divisible by "
                                 f"sequential_same_speaker_samples={sequential_same_speaker_samples}")
        elif pairs is None:
            raise ValueError("mode 'reproduce' needs the list of pairs")
        self.batch_size, self.max_queue_size, self.mode = batch_size, max_queue_size, mode
        self.seq = sequential_same_speaker_samples
        self.collate_fn = collate_fn
        self.ratio = pos_neg_training_batch_ratio
        self.pairs = pairs
        self.fixed_random_seed = fixed_random_seed
        self.random_state = random.getstate()
        self.yield_limit = yield_limit

    def __call__(self, samples: Iterable[SpeakerClassificationDataSample]
                 ) -> Iterator[PairedSpeakerClassificationDataBatch]:
        return self._generate(samples) if self.mode == "generate" else self._reproduce(samples)

    # ------------------------------------------------------------------ training pairs
    def _generate(self, samples):
        if self.fixed_random_seed:
            random.setstate(self.random_state)
        n_pos = round(self.ratio * self.batch_size)
        n_neg = self.batch_size - n_pos
        if not 0 <= n_pos <= self.batch_size:
            raise ValueError(f"pos/neg ratio {self.ratio} gives {n_pos} positives for batches of {self.batch_size}")
        # the reference's threshold: min(batch, floor(queue / batch) * batch) == batch_size once queue >= batch
        threshold = min(self.batch_size, (self.max_queue_size // self.batch_size) * self.batch_size)
        queue: List[SpeakerClassificationDataSample] = []
        yielded = 0
        left_in_run = self.seq
        for s in samples:
            queue.append(s)
            left_in_run -= 1
            if left_in_run > 0:
                continue                       # only look at the queue at the end of a same-speaker run
            left_in_run = self.seq
            if len(queue) < threshold:
                continue
            batch = self.draw_paired_batch(queue, self.batch_size, n_pos, n_neg, self.seq)
            if batch is None:
                raise ValueError("cannot yield batch while data is still being loaded")
            yield self.collate_fn(batch)
            yielded += self.batch_size
            if self.yield_limit is not None and yielded >= self.yield_limit:
                break
        exhausted = False
        while queue:
            if exhausted:
                raise ValueError("queue not empty while out of data")
            batch = self.draw_paired_batch(queue, self.batch_size, n_pos, n_neg, self.seq)
            if batch is None:
                exhausted = True
            else:
                yield self.collate_fn(batch)
                yielded += self.batch_size
            if self.yield_limit is not None and yielded >= self.yield_limit:
                break

    @staticmethod
    def draw_paired_batch(queue: List[SpeakerClassificationDataSample], batch_size: int, n_pos: int, n_neg: int,
                          seq: int) -> Optional[List[PairedSpeakerClassificationDataSample]]:
        """One batch of pairs out of ``queue`` (consumed in place); None (and the queue dropped) when fewer than
        ``batch_size`` samples are left."""
        if len(queue) < batch_size:
            queue.clear()
            return None
        by_speaker: Dict[int, List[SpeakerClassificationDataSample]] = defaultdict(list)
        for s in queue:
            by_speaker[s.ground_truth].append(s)
        if any(len(v) % seq for v in by_speaker.values()):
            raise AssertionError("samples of a speaker must arrive in runs of sequential_same_speaker_samples")
        ids = list(by_speaker)
        counts = [len(by_speaker[i]) for i in ids]
        weights = [2 ** c for c in counts]
        n_speakers = batch_size // seq
        if sum(counts) < batch_size:
            raise ValueError(f"not enough speakers to generate paired batch: queue {len(queue)}, counts {counts}")
        chosen: List[int] = []
        while len(chosen) < n_speakers and ids:
            pick = random.choices(population=ids, weights=weights, k=1)[0]
            at = ids.index(pick)
            chosen.append(pick)
            del ids[at], counts[at], weights[at]
        taken: Dict[int, List[SpeakerClassificationDataSample]] = defaultdict(list)
        for spk in chosen:
            pool = by_speaker[spk]
            for _ in range(seq):
                taken[spk].append(pool.pop(random.randint(0, len(pool) - 1)))

        def seen(pairs, a, b):                 # ORDERED pair (a, b) already present?
            return any(p.primary_key == a.key and p.secondary_key == b.key for p in pairs)

        positives: List[PairedSpeakerClassificationDataSample] = []
        fails = 0
        while len(positives) != n_pos:
            if fails >= 100:
                raise ValueError("too many fails generating positive pairs")
            own = taken[random.choice(chosen)]
            if len(own) < 2:
                fails += 1
                continue
            a, b = random.sample(own, 2)
            if seen(positives, a, b):
                fails += 1
                continue
            positives.append(_make_pair(a, b, 1))
        negatives: List[PairedSpeakerClassificationDataSample] = []
        fails = 0
        while len(negatives) != n_neg:
            if fails >= 100:
                raise ValueError("too many fails generating negative pairs")
            s1, s2 = random.sample(chosen, 2)
            l1, l2 = taken[s1], taken[s2]
            if len(l1) < 1 or len(l2) < 1:
                fails += 1                     # (reference: counted, the draw below still happens)
            a, b = random.choice(l1), random.choice(l2)
            if seen(positives, a, b):          # (reference: checked against the positives only)
                fails += 1
                continue
            negatives.append(_make_pair(a, b, 0))
        for lst in taken.values():
            for s in lst:
                queue.remove(s)
        out = positives + negatives
        random.shuffle(out)
        return out

    # ------------------------------------------------------------------ evaluation pairs
    def _reproduce(self, samples):
        by_key = {s.key: s for s in samples}
        if not by_key:
            return                             # more workers than shards: nothing to do
        batch: List[PairedSpeakerClassificationDataSample] = []
        for pr in self.pairs:
            batch.append(_make_pair(by_key[pr.sample1_id], by_key[pr.sample2_id], 1 if pr.same_speaker else 0))
            if len(batch) == self.batch_size:
                yield self.collate_fn(batch)
                batch = []
        if batch:
            yield self.collate_fn(batch)
