#!/usr/bin/env python3
"""Grouped weight gradients of the training step (two w2v2-base blocks = 8 problems, 216 tiles of 256x256; one block = 4
problems) on each kernel family of csrc/wgrad*.hip (w2v2_tune_wgrad_kernel): 2 = 256x128 ring, 3 = 256x256x32 ring,
4 = 256x256x64 phased.   python3 tools/wgrad_bench.py        FAMILIES=3,4 REPS=20 TRIALS=5 DT=bf16"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"
lp = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float16
M = 66 * 149
Mp = (M + 63) // 64 * 64
H, I = 768, 3072
FAM = {0: "auto", 2: "256x128 ring", 3: "256x256x32 ring", 4: "256x256x64 phased", 5: "phased, 1 of 2 pieces late", 6: "phased, both late"}
fams = [int(x) for x in os.environ.get("FAMILIES", "3,4,5,6").split(",")]
reps, trials = int(os.environ.get("REPS", "20")), int(os.environ.get("TRIALS", "5"))


def mk(c):
    t = torch.zeros(Mp, c, dtype=lp, device=dev)
    t[:M] = (torch.randn(M, c, device=dev) * 0.5).to(lp)
    return t


_bA = torch.randn(66 * 2399, 1536, device=dev).to(lp)
_bB = torch.randn(512, 1536, device=dev).to(lp)
_bC = torch.zeros(66 * 2399, 512, dtype=lp, device=dev)
blocker = ops.Gemm(66 * 2399, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
print(f"# {lp}, tokens {M}, {reps} reps x {trials} trials (median)")
for blocks in (2, 1):
    probs = []
    for _ in range(blocks):
        probs += [(mk(H), mk(I)), (mk(I), mk(H)), (mk(H), mk(H)), (mk(3 * H), mk(H))]
    outs = [(torch.zeros(dy.shape[1], x.shape[1], device=dev), torch.zeros(dy.shape[1], device=dev)) for dy, x in probs]
    wg = ops.WgradGroup([(dy, x, dw, db) for (dy, x), (dw, db) in zip(probs, outs)], M, Mp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    line = []
    ref = None
    for fam in fams:
        ops.lib().w2v2_tune_wgrad_kernel(fam)
        for dw, db in outs:
            dw.zero_(); db.zero_()
        wg()
        torch.cuda.synchronize()
        cur = [(dw.clone(), db.clone()) for dw, db in outs]
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(ref, cur)), f"family {fam} differs"
    for fam in fams:
        ops.lib().w2v2_tune_wgrad_kernel(fam)
        for _ in range(3):
            wg()
        torch.cuda.synchronize()
        ts = []
        for _ in range(trials):
            for _ in range(2):
                blocker()
            e0.record()
            for _ in range(reps):
                wg()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        us = sorted(ts)[len(ts) // 2]
        line.append(f"[{FAM[fam]}] {us:7.1f} us {wg.flops / us / 1e6:6.0f} TF/s")
    ops.lib().w2v2_tune_wgrad_kernel(0)
    print(f"{blocks} block(s), {len(probs)} problems: " + "   ".join(line), flush=True)
