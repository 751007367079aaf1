#!/usr/bin/env python3
"""Which torch (aten) operators still launch device work inside the steady-state training step?

Runs the benchmarked step (configs[1], fp16) under torch.profiler with Python stacks and prints every aten op that
launched a kernel or a device copy, with its call site inside this package -- the list the round-2 review asked to be
empty ("no at::native kernel in the steady-state step")."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from bench import synth_batch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import OneCycle
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import SpeakerTrainer

STEPS = int(os.environ.get("STEPS", "4"))
dev = torch.device("cuda:0")
MODEL, BATCH, NS = os.environ.get("MODEL", "base"), int(os.environ.get("BATCH", "66")), int(os.environ.get("SAMPLES", "48000"))
cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-" + MODEL)      # MODEL=large BATCH=32 SAMPLES=80000: configs[3]
store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994, freeze_cnn=True, embed_dim=2 * cfg.hidden_size)
store.init_weights(seed=20211)
plan = Plan(store, BATCH, NS, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7, pooling="mean+std")
tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=100), layerdrop_seed=1234, mask_seed=7)
wav, label = synth_batch(BATCH, NS, 5994, seed=42133724, device=dev)
for _ in range(3):
    tr.train_step(wav, label)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        tr.train_step(wav, label)
    torch.cuda.synchronize()

sites = Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                     # count the outermost aten op only
    where = next((s for s in (ev.stack or []) if "w2v2_speaker_amd" in s or "bench.py" in s), "?")
    shapes = str(ev.input_shapes)[:60]
    sites[(ev.name, where.strip()[-90:], shapes)] += 1
print(f"# aten ops with device work over {STEPS} steady-state steps (count / steps = per step)")
for (name, where, shapes), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n / STEPS:7.2f}/step  {name:22s} {shapes:60s} {where}")
