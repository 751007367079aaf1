"""Data feed in the reference's shard format: round trip, preprocessors, shuffle-queue batcher."""
import os
import random

import pytest
import torch

from oracle import w2v2_oracle as O
from w2v2_speaker_amd.data import (AudioChunkSelector, BatchProcessor, DeviceFeeder, InputNormalizer2D, ShardDataset,
                                   SpeakerClassificationDataSample, find_shards, iter_shard, read_meta, write_shards)


def _synthetic(n=23, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        frames = 48000 + int(torch.randint(0, 32000, (1,), generator=g))
        out.append((f"id{10000 + i % 5}/yt{i:03d}/{i:05d}", i % 5, 0.1 * torch.randn(1, frames, generator=g) + 0.02))
    return out


@pytest.mark.parametrize("compress", [False, True])
def test_shard_round_trip_and_folder_meta(tmp_path, compress):
    data = _synthetic()
    paths = write_shards(data, str(tmp_path), samples_per_shard=10, compress=compress)
    assert len(paths) == 3 and paths == find_shards(str(tmp_path), "train_shard_*.tar*")
    assert read_meta(str(tmp_path)) == {"num_samples": 23, "num_speakers": 5, "num_shards": 3}
    got = [x for p in paths for x in iter_shard(p)]
    assert [x["__key__"] for x in got] == [k for k, _, _ in data]
    for x, (k, spk, wav) in zip(got, data):
        assert torch.equal(x["wav.pyd"], wav) and x["meta.json"]["speaker_id_idx"] == spk
        assert x["meta.json"]["num_frames"] == wav.shape[-1] and x["meta.json"]["speaker_id"] == k.split("/")[0]


def test_normaliser_matches_oracle_and_selector_lengths():
    wav = 0.3 * torch.randn(1, 70000) + 0.1
    x, mean, std = InputNormalizer2D.normalize(wav, channel_wise=False)
    assert torch.allclose(x[0], O.normalise_waveform(wav[0]), atol=1e-6)
    s = SpeakerClassificationDataSample("k", 3, wav)
    random.seed(5)
    out = AudioChunkSelector("random", 3.0).process(s)
    assert out.network_input.shape == (1, 48000)
    random.seed(5)
    start = random.randint(0, 70000 - 48000 - 1)                       # the reference's own draw
    assert torch.equal(out.network_input, wav[..., start:start + 48000])
    many = AudioChunkSelector("contiguous", 1.0).process(SpeakerClassificationDataSample("k", 3, wav))
    assert len(many) == 4 and many[2].key == "k/chunk2" and all(m.network_input.shape[-1] == 16000 for m in many)
    short = AudioChunkSelector("random", 3.0).process(SpeakerClassificationDataSample("k", 3, wav[..., :1000]))
    assert short.network_input.shape[-1] == 1000


def test_batch_processor_sees_every_sample_once_and_is_seed_deterministic():
    samples = [SpeakerClassificationDataSample(f"k{i}", i % 7, torch.full((1, 8), float(i))) for i in range(50)]
    with pytest.raises(ValueError):
        BatchProcessor(8, 4)
    random.seed(1)
    a = list(BatchProcessor(8, 16)(iter(samples)))
    random.seed(1)
    b = list(BatchProcessor(8, 16)(iter(samples)))
    assert [x.keys for x in a] == [x.keys for x in b]
    keys = [k for x in a for k in x.keys]
    assert sorted(keys) == sorted(s.key for s in samples)
    assert all(x.batch_size == 8 for x in a[:-1]) and a[0].network_input.shape == (8, 1, 8)
    assert a[0].ground_truth.dtype == torch.int64
    assert keys != [s.key for s in samples]                             # shuffled


def test_shard_dataset_end_to_end_and_feeder_on_cpu(tmp_path):
    paths = write_shards(_synthetic(30), str(tmp_path), samples_per_shard=16)
    random.seed(3)
    ds = ShardDataset(paths, batch_size=6, queue_size=12)
    batches = list(DeviceFeeder(ds, "cpu", depth=2))
    assert sum(b.batch_size for b in batches) == 30
    b0 = batches[0]
    assert b0.network_input.shape == (6, 1, 48000) and b0.ground_truth.shape == (6,)
    assert abs(float(b0.network_input.mean())) < 0.05          # chunk of a normalised utterance


def test_fbank_shapes_and_tone_localisation():
    """x-vector / ECAPA front-end (config/data/pipeline/xvector_pipeline.yaml): 3 s -> [301, 40] log-mel frames; a pure
    tone lights up the filter whose centre is closest; then channel-wise normalisation as in the pipeline."""
    import math
    from w2v2_speaker_amd.data import Fbank, FilterBank, InputNormalizer2D
    fb = Fbank(40)
    t = torch.arange(48000) / 16000.0
    for f in (300.0, 1000.0, 3000.0):
        feat = fb(torch.sin(2 * math.pi * f * t)[None])
        assert feat.shape == (301, 40) and torch.isfinite(feat).all()
        mel = 2595.0 * math.log10(1 + f / 700.0)
        centres = torch.linspace(0, 2595.0 * math.log10(1 + 8000 / 700.0), 42)[1:-1]
        assert abs(int(feat[50:250].mean(0).argmax()) - int((centres - mel).abs().argmin())) <= 1
        assert float(feat.max() - feat.min()) <= 80.0 + 1e-3
    s = SpeakerClassificationDataSample("k", 0, 0.1 * torch.randn(1, 48000))
    s = FilterBank(40).process(s)
    s = InputNormalizer2D(normalize_over_channels=True).process(s)
    assert s.network_input.shape == (301, 40) and float(s.network_input.mean(0).abs().max()) < 1e-4



# ------------------------------------------------------------------------------------------------ pair batcher
def _paired_stream(n_speakers, per_speaker, seq, seed):
    import random as _r
    from w2v2_speaker_amd.data import SpeakerClassificationDataSample
    rng = _r.Random(seed)
    runs = [(spk, r) for spk in range(n_speakers) for r in range(per_speaker // seq)]
    rng.shuffle(runs)
    for spk, r in runs:
        for j in range(seq):
            yield SpeakerClassificationDataSample(key=f"id{spk:03d}/vid{r:02d}/{j:05d}", ground_truth=spk,
                                                  network_input=torch.full((1, 8), float(spk * 1000 + r * 10 + j)))


def test_paired_batch_processor_matches_reference_goldens():
    """tests/golden/data_pipeline.json was produced by the reference's own PairedBatchProcessor
    (tests/golden/make_data_goldens.py): same seed -> the same ordered pairs in the same batches."""
    import json
    import os
    import random
    from w2v2_speaker_amd.data.paired import EvaluationPair, PairedBatchProcessor
    with open(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        if case["mode"] == "generate":
            proc = PairedBatchProcessor(case["batch_size"], case["max_queue_size"], "generate", case["seq"],
                                        pos_neg_training_batch_ratio=case["ratio"])
            random.seed(case["seed"])
            src = _paired_stream(case["n_speakers"], case["per_speaker"], case["seq"], case["seed"])
        else:
            pairs = [EvaluationPair(*p) for p in case["pairs"]]
            proc = PairedBatchProcessor(case["batch_size"], 8, "reproduce", 1, pairs=pairs)
            src = _paired_stream(case["n_speakers"], case["per_speaker"], case["seq"], case["seed"])
        got = [[[p, s, int(g)] for p, s, g in zip(b.primary_keys, b.secondary_keys, b.ground_truth.tolist())]
               for b in proc(src)]
        assert got == case["batches"], case["mode"]
        for b in proc(_paired_stream(case["n_speakers"], case["per_speaker"], case["seq"], case["seed"])) \
                if case["mode"] == "reproduce" else []:
            assert b.primary_network_input.shape == (b.batch_size, 8)


def test_paired_batch_processor_properties_and_errors(tmp_path):
    import random
    from w2v2_speaker_amd.data.paired import PairedBatchProcessor, read_test_pairs_file
    with pytest.raises(ValueError):
        PairedBatchProcessor(8, 4, "generate", 2, pos_neg_training_batch_ratio=0.5)        # queue < batch
    with pytest.raises(ValueError):
        PairedBatchProcessor(8, 16, "generate", 3, pos_neg_training_batch_ratio=0.5)       # batch % seq
    with pytest.raises(ValueError):
        PairedBatchProcessor(8, 16, "generate", 2)                                         # no ratio
    with pytest.raises(ValueError):
        PairedBatchProcessor(8, 16, "reproduce", 2)                                        # no pairs
    with pytest.raises(ValueError):
        PairedBatchProcessor(8, 16, "shuffle", 2)
    random.seed(0)
    proc = PairedBatchProcessor(8, 16, "generate", 2, pos_neg_training_batch_ratio=0.5, yield_limit=24)
    batches = list(proc(_paired_stream(40, 2, 2, 0)))
    assert len(batches) == 3                                                               # yield_limit / batch
    for b in batches:
        assert b.batch_size == 8 and int(b.ground_truth.sum()) == 4
        for p, s, g in zip(b.primary_keys, b.secondary_keys, b.ground_truth.tolist()):
            assert (p.split("/")[0] == s.split("/")[0]) == bool(g) and p != s
    f = tmp_path / "trials.txt"
    f.write_text("1 a/b/c.wav a/d/e.wav\nbroken line\n0 a/b/c.wav x/y/z.wav\n")
    got = list(read_test_pairs_file(f))
    assert [(p.same_speaker, p.sample1_id) for p in got] == [(True, "a/b/c.wav"), (False, "a/b/c.wav")]


def test_batch_processor_and_chunk_selector_match_reference_goldens():
    """Same golden file: the reference's BatchProcessor (shuffle queue) and AudioChunkSelector were run on seeded
    inputs; ours must pop / crop identically under the same ``random.seed``."""
    import json
    import os
    import random
    from w2v2_speaker_amd.data import AudioChunkSelector, BatchProcessor, SpeakerClassificationDataSample
    with open(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.json")) as f:
        gold = json.load(f)
    for c in gold["batch_processor"]:
        random.seed(c["seed"])
        bp = BatchProcessor(c["max_batch_size"], c["max_queue_size"])
        got = [list(b.keys) for b in bp(_paired_stream(c["n_speakers"], 2, 2, c["seed"]))]
        assert got == c["batches"], c
    for c in gold["chunk_selector"]:
        random.seed(c["seed"])
        sel = AudioChunkSelector(c["strategy"], c["sec"])
        for want in c["results"]:
            smp = SpeakerClassificationDataSample(key="k", ground_truth=0,
                                                  network_input=torch.arange(c["n"], dtype=torch.float32).view(1, -1))
            if want == "ValueError":
                with pytest.raises(ValueError):
                    sel.process(smp)
                continue
            r = sel.process(smp)
            r = r if isinstance(r, list) else [r]
            assert [[x.key, int(x.network_input[0, 0]), int(x.network_input.shape[-1])] for x in r] == want, c
