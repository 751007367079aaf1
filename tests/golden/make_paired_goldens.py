"""Golden vectors for the pair batcher: runs the REFERENCE's PairedBatchProcessor (authoring container only; needs
/root/reference) on a seeded synthetic sample stream and records which ordered pairs each batch holds.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_paired_goldens.py
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
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "paired_batcher.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0)
print("wrote", path, [len(c["batches"]) for c in out["cases"]])
