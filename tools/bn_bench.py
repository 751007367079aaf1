#!/usr/bin/env python3
"""Stand-alone timing of the BatchNorm kernel pairs of the ECAPA-TDNN step (f32, M = 66 x 300 rows) by kernel, with the
bytes each one must move: partial (statistics) and apply, forward and backward, for the three geometries of the step --
a 128-channel Res2Net slice of a 1024-wide tensor, a full 1024-wide tensor, the 3072-wide MFA tensor.

    python tools/bn_bench.py        (under rocprofv3 --kernel-trace --stats for the per-kernel split)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops as o

dev = "cuda"
M = 66 * 300
reps = int(os.environ.get("REPS", "30"))
flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GB: evicts L2 / MALL between launches


def timed(fn):
    ts = []
    for _ in range(reps):
        flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print(f"# M = {M}, f32, median of {reps} launches with a 1 GB flush in between; us per kernel PAIR (events around both launches)")
print("# geometry                     fwd pair us   GB/s (3 x tensor)   bwd pair us   GB/s (5 x tensor)")
for name, C, ld in (("128-ch slice of 1024", 128, 1024), ("1024 wide", 1024, 1024), ("3072 wide", 3072, 3072)):
    a = torch.randn(M, ld, device=dev)
    y = torch.empty(M, ld, device=dev)
    dy = torch.randn(M, ld, device=dev)
    da = torch.empty(M, ld, device=dev)
    work = o.bn_workspace(M, C, dev)
    mr = torch.empty(2 * C, device=dev)
    running = torch.zeros(2 * C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    cs = torch.empty(o.bn_colsum_rows(M, C), C, device=dev)
    f = timed(lambda: o.bn_fwd(a, ld, work, mr, running, gamma, beta, y, ld, M, C, 1e-5, 0.1, True, True))
    b = timed(lambda: o.bn_bwd(dy, ld, a, ld, mr, gamma, work, dg, db, da, ld, M, C, True, cs))
    t = M * C * 4 / 1e3          # KB -> us * GB/s
    ev = timed(lambda: o.bn_fwd(a, ld, work, mr, running + 1.0, gamma, beta, y, ld, M, C, 1e-5, 0.1, True, False))
    print(f"{name + ': eval apply only':28s} {ev:12.1f} {2 * t / ev:18.0f}    (one kernel, 2 x tensor, no statistics fold)")
    if C == ld:                   # what the part streams, cold, on the same tensors: a device copy (2 x tensor) and a + dy -> y (3 x)
        cp = timed(lambda: y.copy_(a))
        ad = timed(lambda: torch.add(a, dy, out=y))
        print(f"{name + ': copy / add':28s} {cp:12.1f} {2 * t / cp:18.0f} {ad:13.1f} {3 * t / ad:18.0f}")
    print(f"{name:28s} {f:12.1f} {3 * t / f:18.0f} {b:13.1f} {5 * t / b:18.0f}")
