#!/usr/bin/env python3
"""Time attribution of the 256x128 ring GEMM (the dominant kernel of the step): the products of one transformer block that run on
it, stand-alone, with parts of the K loop compiled out by w2v2_tune_gemm_ring_debug (garbage results, timing only):
    16 the attribution kernel with nothing removed (its own baseline)   + 1 no LDS-DMA pieces in the
    steady-state loop   + 2 no barrier   + 4 no vmcnt wait   + 8 no fragment reads   + 32 no epilogue (combinations add); 0 = the product kernel
Median of ROUNDS interleaved rounds of 5 launches.   python3 tools/ring_attrib.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import _lib, ops

lib = _lib.load()
dev = "cuda"
M = 66 * 149
shapes = [("out-proj (single term)", M, 768, 768), ("QKV (single term)", M, 2304, 768), ("FFN2 / dX1", M, 768, 3072),
          ("dX (K = 2304)", M, 768, 2304), ("one K tile (N = 768)", M, 768, 64), ("two K tiles (N = 768)", M, 768, 128)]
codes = [int(c) for c in os.environ.get("CODES", "0 16 17 18 20 24 25 30 31 48 63").split()]
rounds = int(os.environ.get("ROUNDS", "5"))
print(f"{'product':24s} {'M':>5} {'N':>5} {'K':>5} " + " ".join(f"{c:>7d}" for c in codes) + "   us (median); code 31 = bare MFMA loop + prologue + epilogue")
for name, m, n, k in shapes:
    A = (torch.randn(m, k, device=dev) * 0.1).half()
    B = (torch.randn(n, k, device=dev) * 0.1).half()
    C = torch.zeros(m, n, device=dev, dtype=torch.float16)
    gm = ops.Gemm(m, n, k, A, B, C, lda=k, ldb=k, ldc=n)
    assert gm.kernel_name == "gemm16_ring_256x128_kernel", gm.kernel_name
    t = {c: [] for c in codes}
    for r in range(rounds + 1):
        for c in codes:
            lib.w2v2_tune_gemm_ring_debug(c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gm()
            e0.record()
            for _ in range(5):
                gm()
            e1.record()
            torch.cuda.synchronize()
            if r:
                t[c].append(e0.elapsed_time(e1) * 200.0)
    lib.w2v2_tune_gemm_ring_debug(0)
    row = [statistics.median(t[c]) for c in codes]
    print(f"{name:24s} {m:5d} {n:5d} {k:5d} " + " ".join(f"{u:7.1f}" for u in row) +
          f"   {2.0 * m * n * k / row[0] / 1e6:6.0f} TFLOP/s", flush=True)
