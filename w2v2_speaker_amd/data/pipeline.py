"""Preprocessors and the shuffle-queue batcher of the reference's training pipeline
(``config/data/pipeline/wav2vec_base_pipeline.yaml``: normalizer -> selector_train)."""
from __future__ import annotations

import dataclasses
import random
from typing import Callable, Iterable, Iterator, List, Optional, Union

import torch

from ..lightning_modules.speaker.wav2vec2_fc import SpeakerClassificationDataBatch


@dataclasses.dataclass
class SpeakerClassificationDataSample:
    """ref: src/data/modules/speaker/training_batch_speaker.py:27-41."""
    key: str
    ground_truth: int
    network_input: torch.Tensor
    side_info: Optional[object] = None


def to_sample(x: dict) -> SpeakerClassificationDataSample:
    """ref: voxceleb.py:562-583 (NaN check included)."""
    wav = x["wav.pyd"]
    if torch.any(torch.isnan(wav)):
        raise ValueError(f"NaN value in audio sample of key={x['__key__']}")
    return SpeakerClassificationDataSample(key=x["__key__"], ground_truth=x["meta.json"]["speaker_id_idx"],
                                           network_input=wav)


class InputNormalizer2D:
    """ref: src/data/preprocess/input_normalisation.py:44-75: (x - mean) / (std_unbiased + 1e-5), over the whole
    2-D input (``normalize_over_channels: false`` in the wav2vec2 pipelines) or per feature column."""

    def __init__(self, normalize_over_channels: bool = True):
        self.channel_wise = normalize_over_channels

    @staticmethod
    def normalize(x: torch.Tensor, channel_wise: bool):
        if x.dim() != 2:
            raise ValueError("expect to normalize over 2D input")
        std, mean = torch.std_mean(x, dim=0) if channel_wise else torch.std_mean(x)
        return (x - mean) / (std + 1e-5), mean, std

    def process(self, sample: SpeakerClassificationDataSample):
        sample.network_input = self.normalize(sample.network_input, self.channel_wise)[0]
        return sample


class AudioChunkSelector:
    """ref: src/data/preprocess/random_chunks.py:53-165.  Same ``random`` module calls as the reference, so a seeded
    ``random.seed`` reproduces its crops (note ``randint(0, n - chunk - 1)``: the last start is never drawn)."""

    def __init__(self, selection_strategy: str, desired_chunk_length_sec: float, sample_rate: int = 16000):
        fns = {"start": self._start, "end": self._end, "random": self._random,
               "random_contiguous": self._random_contiguous, "contiguous": self._contiguous}
        if selection_strategy not in fns:
            raise ValueError(f"unknown selection strategy {selection_strategy}")
        self.fn = fns[selection_strategy]
        self.chunk_size = round(sample_rate * desired_chunk_length_sec)

    def process(self, sample: SpeakerClassificationDataSample
                ) -> Union[SpeakerClassificationDataSample, List[SpeakerClassificationDataSample]]:
        chunks = list(self.fn(sample.network_input))
        if len(chunks) == 1:
            sample.network_input = chunks[0]
            return sample
        if not chunks:
            raise ValueError("unable to select at least one chunk")
        return [SpeakerClassificationDataSample(key=sample.key + f"/chunk{i}", network_input=c,
                                                ground_truth=sample.ground_truth, side_info=sample.side_info)
                for i, c in enumerate(chunks)]

    def _start(self, w):
        yield w[..., : self.chunk_size]

    def _end(self, w):
        yield w[..., -self.chunk_size:]

    def _random(self, w):
        n = w.shape[-1]
        if self.chunk_size > n:
            yield w[..., :]
        else:
            start = random.randint(0, n - self.chunk_size - 1)
            yield w[..., start:start + self.chunk_size]

    def _random_contiguous(self, w):
        k = random.randint(0, w.shape[-1] // self.chunk_size - 1)
        yield w[..., k * self.chunk_size:(k + 1) * self.chunk_size]

    def _contiguous(self, w):
        for k in range(w.shape[-1] // self.chunk_size):
            yield w[..., k * self.chunk_size:(k + 1) * self.chunk_size]


def default_collate_fn(lst: List[SpeakerClassificationDataSample]) -> SpeakerClassificationDataBatch:
    """ref: training_batch_speaker.py:78-95 (``default_collate`` = stack)."""
    return SpeakerClassificationDataBatch(
        batch_size=len(lst), keys=[s.key for s in lst], network_input=torch.stack([s.network_input for s in lst]),
        ground_truth=torch.tensor([int(s.ground_truth) for s in lst], dtype=torch.int64),
        side_info={s.key: s.side_info for s in lst})


class BatchProcessor:
    """ref: voxceleb.py:829-886: samples accumulate in a queue of ``max_queue_size``; once full, batches are drawn
    by popping uniformly random queue positions (``random.randint``); the tail is drained at the end."""

    def __init__(self, max_batch_size: int, max_queue_size: int,
                 collate_fn: Callable[[List[SpeakerClassificationDataSample]], SpeakerClassificationDataBatch]
                 = default_collate_fn):
        if max_batch_size <= 0:
            raise ValueError("max_batch_size needs to be a positive integer")
        if max_queue_size <= 0 or max_queue_size < max_batch_size:
            raise ValueError(f"queue size needs to be >= max(1, max_batch_size={max_batch_size}), "
                             f"while given value is {max_queue_size}")
        self.max_batch_size, self.max_queue_size, self.collate_fn = max_batch_size, max_queue_size, collate_fn
        self.queue: List[SpeakerClassificationDataSample] = []

    def __call__(self, sample_iterator: Iterable[SpeakerClassificationDataSample]
                 ) -> Iterator[SpeakerClassificationDataBatch]:
        self.queue.clear()
        for sample in sample_iterator:
            if not isinstance(sample, SpeakerClassificationDataSample):
                raise ValueError(f"batch is expected to be of type {SpeakerClassificationDataSample}")
            self.queue.append(sample)
            if len(self.queue) >= self.max_queue_size:
                yield self._get_batch()
        while len(self.queue) >= 1:
            yield self._get_batch()
        self.queue.clear()

    def _get_batch(self) -> SpeakerClassificationDataBatch:
        if len(self.queue) == 0:
            raise ValueError("cannot get a batch without any samples")
        batch = []
        while len(batch) < self.max_batch_size and len(self.queue) >= 1:
            batch.append(self.queue.pop(random.randint(0, len(self.queue) - 1)))
        return self.collate_fn(batch)
