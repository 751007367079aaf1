#!/usr/bin/env python3
"""Exact-f32 GEMM: a few ECAPA-sized products under forced kernels / tiles (w2v2_tune_gemm_f32_tile codes), HIP-event
timed: ROUNDS interleaved rounds of 5 launches per code, the median per code is printed.
    CODES="0 14 15 215 415" python3 tools/f32_gemm_bench.py
codes: 1..4 register-staged tiles; 11..15 LDS-DMA fi = 1..5; +100 XCD-contiguous tiles;
+200 no DMA in the loop, +400 no barrier, +800 no vmcnt wait (debug variants: garbage results, timing only).
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import _lib, ops

lib = _lib.load()
dev = "cuda"
shapes = [(19800, 1024, 1024, False, False), (19800, 3072, 3072, False, False), (19800, 1024, 1024, False, True),
          (19800, 3072, 3072, False, True), (1024, 1024, 19800, True, True), (3072, 3072, 19800, True, True),
          (19800, 128, 384, False, False), (19800, 384, 128, False, True), (128, 384, 19800, True, True)]
if os.environ.get("SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split()]
codes = [int(c) for c in os.environ.get("CODES", "0 1 14 15 115").split()]
rounds = int(os.environ.get("ROUNDS", "5"))
print(f"{'M':>6} {'N':>5} {'K':>6} tA tB split " + " ".join(f"{c:>8d}" for c in codes) + "   (us, median; TFLOP/s of the best)")
for (M, N, K, ta, tb) in shapes:
    A = torch.randn(K, M, device=dev) if ta else torch.randn(M, K, device=dev)
    B = torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)
    C = torch.zeros(M, N, device=dev)
    split = (32 if M * N < 1 << 18 else 8) if K > 8192 else 1

    def run():
        ops.gemm(M, N, K, A, B, C, lda=M if ta else K, ldb=N if tb else K, ldc=N, transA=ta, transB=tb, alpha=1.0,
                 split_k=split)
    t = {c: [] for c in codes}
    for r in range(rounds + 1):
        for c in codes:
            lib.w2v2_tune_gemm_f32_tile(c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run()
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r:
                t[c].append(e0.elapsed_time(e1) * 200.0)
    lib.w2v2_tune_gemm_f32_tile(0)
    row = [statistics.median(t[c]) for c in codes]
    best = min(row)
    print(f"{M:6d} {N:5d} {K:6d} {int(ta):2d} {int(tb):2d} {split:5d} " + " ".join(f"{u:8.1f}" for u in row) +
          f"   {2.0 * M * N * K / best / 1e6:6.1f}", flush=True)
