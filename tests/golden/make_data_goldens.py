"""Golden vectors for the host-side data pipeline: runs the REFERENCE's PairedBatchProcessor, BatchProcessor and
AudioChunkSelector (authoring container only; needs /root/reference) on seeded synthetic sample streams and records
which (ordered pairs of) keys each batch holds and which crops are taken.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_data_goldens.py
"""
import json
import os
import random
import sys
import types

import torch

REF = "/root/reference"
sys.path.insert(0, REF)
torch.cuda.Device = torch.device
pl = types.ModuleType("pytorch_lightning")
pl.LightningModule, pl.LightningDataModule, pl.Trainer = torch.nn.Module, object, object
sys.modules["pytorch_lightning"] = pl
for n in ("hurry", "hurry.filesize", "torchaudio", "webdataset", "seaborn", "yaspin", "dotenv", "omegaconf", "hydra",
          "hydra.utils"):
    sys.modules.setdefault(n, types.ModuleType(n))
sys.modules["hurry.filesize"].size = lambda *a, **k: ""
sys.modules["omegaconf"].DictConfig = dict
sys.modules["omegaconf"].OmegaConf = object
sys.modules["omegaconf"].MISSING = None

from src.data.modules.speaker.voxceleb import PairedBatchProcessor  # noqa: E402
from src.data.modules.speaker.training_batch_speaker import (  # noqa: E402
    PairedSpeakerClassificationDataBatch, SpeakerClassificationDataSample)
from src.evaluation.speaker.speaker_recognition_evaluator import EvaluationPair  # noqa: E402


def stream(n_speakers, per_speaker, seq, seed):
    """speaker-grouped runs of `seq` samples, as the reference's shard writer lays them out"""
    rng = random.Random(seed)
    runs = [(spk, r) for spk in range(n_speakers) for r in range(per_speaker // seq)]
    rng.shuffle(runs)
    for spk, r in runs:
        for j in range(seq):
            key = f"id{spk:03d}/vid{r:02d}/{j:05d}"
            yield SpeakerClassificationDataSample(key=key, ground_truth=spk,
                                                  network_input=torch.full((1, 8), float(spk * 1000 + r * 10 + j)),
                                                  side_info=None)


def record(batches):
    return [[(p, s, int(g)) for p, s, g in zip(b.primary_keys, b.secondary_keys, b.ground_truth.tolist())]
            for b in batches]


out = {"cases": []}
for case in [dict(batch_size=8, max_queue_size=32, seq=2, ratio=0.5, n_speakers=12, per_speaker=6, seed=11),
             dict(batch_size=12, max_queue_size=24, seq=4, ratio=0.25, n_speakers=20, per_speaker=4, seed=5),
             dict(batch_size=6, max_queue_size=64, seq=3, ratio=0.67, n_speakers=14, per_speaker=3, seed=3)]:
    proc = PairedBatchProcessor(batch_size=case["batch_size"], max_queue_size=case["max_queue_size"], mode="generate",
                                sequential_same_speaker_samples=case["seq"],
                                collate_fn=PairedSpeakerClassificationDataBatch.default_collate_fn,
                                pos_neg_training_batch_ratio=case["ratio"])
    random.seed(case["seed"])
    got = record(proc(stream(case["n_speakers"], case["per_speaker"], case["seq"], case["seed"])))
    out["cases"].append({**case, "mode": "generate", "batches": got})

samples = list(stream(5, 4, 2, 1))
rng = random.Random(9)
pairs = []
for _ in range(11):
    a, b = rng.sample(samples, 2)
    pairs.append(EvaluationPair(a.ground_truth == b.ground_truth, a.key, b.key))
proc = PairedBatchProcessor(batch_size=4, max_queue_size=8, mode="reproduce", sequential_same_speaker_samples=1,
                            collate_fn=PairedSpeakerClassificationDataBatch.default_collate_fn, pairs=pairs)
out["cases"].append({"mode": "reproduce", "batch_size": 4, "pairs": [[bool(p.same_speaker), p.sample1_id, p.sample2_id] for p in pairs],
                     "n_speakers": 5, "per_speaker": 4, "seq": 2, "seed": 1, "batches": record(proc(iter(samples)))})
# ---- single-sample shuffle queue (voxceleb.py:829-886) and the chunk selector (random_chunks.py)
from src.data.modules.speaker.voxceleb import BatchProcessor  # noqa: E402
from src.data.modules.speaker.training_batch_speaker import SpeakerClassificationDataBatch  # noqa: E402
from src.data.preprocess.random_chunks import AudioChunkSelector  # noqa: E402

out["batch_processor"] = []
for bs, qs, n_spk, seed in [(4, 10, 9, 2), (5, 5, 4, 8), (3, 64, 5, 1)]:
    random.seed(seed)
    bp = BatchProcessor(bs, qs, SpeakerClassificationDataBatch.default_collate_fn)
    got = [list(b.keys) for b in bp(stream(n_spk, 2, 2, seed))]
    out["batch_processor"].append({"max_batch_size": bs, "max_queue_size": qs, "n_speakers": n_spk, "seed": seed,
                                   "batches": got})
out["chunk_selector"] = []
for strategy in ["start", "end", "random", "random_contiguous", "contiguous"]:
    for n, sec in [(1000, 0.02), (321, 0.02), (200, 0.02)]:       # chunk = 320 samples at 16 kHz
        random.seed(n + len(strategy))
        sel = AudioChunkSelector(strategy, sec)
        res = []
        for rep in range(3):
            smp = SpeakerClassificationDataSample(key="k", ground_truth=0,
                                                  network_input=torch.arange(n, dtype=torch.float32).view(1, n),
                                                  side_info=None)
            try:
                r = sel.process(smp)
                r = r if isinstance(r, list) else [r]
                res.append([[x.key, int(x.network_input[0, 0]), int(x.network_input.shape[-1])] for x in r])
            except ValueError as e:
                res.append("ValueError")
        out["chunk_selector"].append({"strategy": strategy, "n": n, "sec": sec, "seed": n + len(strategy),
                                      "results": res})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data_pipeline.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0)
print("wrote", path, [len(c["batches"]) for c in out["cases"]])
