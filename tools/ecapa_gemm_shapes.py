#!/usr/bin/env python3
"""Per-shape table of the matrix products of ONE ECAPA-TDNN training step (configs[4], f32 by default): every distinct
descriptor the step launches, timed stand-alone (HIP events, 10 launches) -- the ECAPA counterpart of
tools/gemm_shapes.py.    python3 tools/ecapa_gemm_shapes.py [bf16]"""
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
from w2v2_speaker_amd.ecapa import EcapaConfig, EcapaPlan, EcapaStore, EcapaTrainer
from w2v2_speaker_amd.optim.schedule import OneCycle

dev = torch.device("cuda:0")
dt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32
cfg = EcapaConfig()
st = EcapaStore(cfg, dev, dt, num_speakers=5994)
st.init_weights(1)
plan = EcapaPlan(st, 66, 300, train=True)
tr = EcapaTrainer(st, plan, OneCycle(max_lr=1e-3, total_steps=100), None)
g = torch.Generator().manual_seed(0)
feats = torch.randn(66, 300, cfg.input_mel_coefficients, generator=g).to(dev)
labels = torch.randint(0, 5994, (66,), generator=g).to(dev)
for _ in range(2):
    tr.train_step(feats, labels)
torch.cuda.synchronize()
seen = OrderedDict()
orig = ops.Gemm.__call__


def rec(self):
    seen.setdefault(id(self), [self, 0])[1] += 1
    return orig(self)


# F32_TILE=1..4 forces the f32 tile (1 = 128x128, 2 = 64x128, 3 = 128x64, 4 = 64x64) for every product (w2v2_tune_gemm_f32_tile)
if os.environ.get("F32_TILE"):
    from w2v2_speaker_amd import _lib
    _lib.load().w2v2_tune_gemm_f32_tile(int(os.environ["F32_TILE"]))
ops.Gemm.__call__ = rec
tr.train_step(feats, labels)
torch.cuda.synchronize()
ops.Gemm.__call__ = orig
rows = {}
for gm, n in seen.values():
    d = gm.desc
    key = (d.M, d.N, d.K, d.batch, int(d.A.trans), int(d.B.trans), d.split_k, d.epilogue, int(d.accumulate))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        orig(gm)
    e0.record()
    for _ in range(10):
        orig(gm)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    r = rows.setdefault(key, [0, 0.0, gm.flops, gm.kernel_name])
    r[0] += n
    r[1] += us * n
tot = sum(r[1] for r in rows.values())
print(f"# ECAPA-TDNN step, {dt}: {sum(r[0] for r in rows.values())} products, {tot / 1e3:.2f} ms stand-alone")
print(f"{'M':>7} {'N':>6} {'K':>7} {'bat':>4} {'tA':>2} {'tB':>2} {'splK':>4} {'epi':>3} {'acc':>3} {'n':>4} {'us each':>9} {'ms':>7} {'TF/s':>7}  kernel")
for key, (n, us, fl, kn) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:7d} {key[1]:6d} {key[2]:7d} {key[3]:4d} {key[4]:2d} {key[5]:2d} {key[6]:4d} {key[7]:3d} {key[8]:3d} {n:4d} "
          f"{us / n:9.1f} {us / 1e3:7.2f} {fl * n / us / 1e6:7.1f}  {kn}")
