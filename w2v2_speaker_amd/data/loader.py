"""Shards -> samples -> preprocessors -> shuffle-queue batches (ref: voxceleb.py:418-468), and the host->HBM feeder.

At 4 000+ utterances/s a step consumes a [66, 1, 48000] f32 batch (12.7 MB) every ~14 ms.  ``DeviceFeeder`` runs the
host pipeline in a background thread, stages batches in pinned memory and uploads them on a side HIP stream, so the
copy (0.2 ms at PCIe rates) and the Python work overlap the previous step's kernels (the reference relies on 5
DataLoader workers for the same purpose, config/data/dataloader)."""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, List, Optional, Sequence

import torch

from .pipeline import (AudioChunkSelector, BatchProcessor, InputNormalizer2D, default_collate_fn, to_sample)
from .shards import iter_shard


class ShardDataset:
    """One pass over the shards: decode -> SpeakerClassificationDataSample -> preprocessors (each may return one
    sample or a list, ref ``_pipe_preprocessors`` :586-600) -> BatchProcessor."""

    def __init__(self, shard_paths: Sequence[str], batch_size: int, queue_size: int = 1024,
                 preprocessors: Optional[List] = None, collate_fn=default_collate_fn):
        self.shard_paths = list(shard_paths)
        self.preprocessors = preprocessors if preprocessors is not None else self.train_pipeline()
        self.batcher = BatchProcessor(batch_size, max(queue_size, batch_size), collate_fn)

    @staticmethod
    def train_pipeline(chunk_seconds: float = 3.0):
        """config/data/pipeline/wav2vec_base_pipeline.yaml: normalizer (whole utterance) -> random 3 s chunk."""
        return [InputNormalizer2D(normalize_over_channels=False), AudioChunkSelector("random", chunk_seconds)]

    @staticmethod
    def val_pipeline(chunk_seconds: float = 3.0):
        return [InputNormalizer2D(normalize_over_channels=False), AudioChunkSelector("start", chunk_seconds)]

    def samples(self):
        for path in self.shard_paths:
            for x in iter_shard(path):
                items = [to_sample(x)]
                for p in self.preprocessors:
                    nxt = []
                    for s in items:
                        r = p.process(s)
                        nxt.extend(r if isinstance(r, list) else [r])
                    items = nxt
                yield from items

    def __iter__(self):
        return self.batcher(self.samples())


class DeviceFeeder:
    """Iterate device-resident batches; host pipeline + H2D copy run ahead of the consumer by ``depth`` batches."""

    def __init__(self, batches: Iterable, device, depth: int = 3):
        self.device = torch.device(device)
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._run, args=(iter(batches),), daemon=True)
        self._thread.start()

    def _run(self, it) -> None:
        try:
            for b in it:
                if self._stream is None:
                    self._q.put((b.to(self.device), None))
                    continue
                x, y = b.network_input.pin_memory(), b.ground_truth.pin_memory()
                with torch.cuda.stream(self._stream):
                    dx, dy = x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self._q.put((type(b)(b.batch_size, b.keys, dx, dy, b.side_info), ev))
        except BaseException as e:          # surfaced on the consumer side
            self._err = e
        finally:
            self._q.put(None)

    def __iter__(self) -> Iterator:
        while True:
            item = self._q.get()
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            batch, ev = item
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)      # no host sync
                batch.network_input.record_stream(torch.cuda.current_stream(self.device))
                batch.ground_truth.record_stream(torch.cuda.current_stream(self.device))
            yield batch
