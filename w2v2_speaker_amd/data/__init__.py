"""Data feed in the reference's on-disk format: webdataset-style tar shards of pickled waveform
tensors + json metadata, the shuffle-queue batcher and the normalise / 3 s-crop preprocessors, plus a prefetching
host->HBM feeder so a 4 000+ utterances/s training step is not starved."""
from .pipeline import (AudioChunkSelector, BatchProcessor, InputNormalizer2D, SpeakerClassificationDataSample,
                       default_collate_fn)
from .shards import find_shards, iter_shard, read_meta, write_shards
from .loader import DeviceFeeder, ShardDataset
from .fbank import Fbank, FilterBank
from .paired import (EvaluationPair, PairedBatchProcessor, PairedSpeakerClassificationDataSample,
                     paired_default_collate_fn, read_test_pairs_file)

__all__ = ["AudioChunkSelector", "BatchProcessor", "InputNormalizer2D", "SpeakerClassificationDataSample",
           "default_collate_fn", "find_shards", "iter_shard", "read_meta", "write_shards", "DeviceFeeder",
           "ShardDataset", "Fbank", "FilterBank", "EvaluationPair", "PairedBatchProcessor",
           "PairedSpeakerClassificationDataSample", "paired_default_collate_fn", "read_test_pairs_file"]
