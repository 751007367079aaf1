#!/usr/bin/env python3
"""Repeat the step's GEMM products many times on fixed operands and compare every output with the first one bit for bit: the
LDS-DMA pieces of the ring / phased / f32 kernels are inline assembly that the compiler does not order (round 6), so a missing
counted wait would show up here as an intermittent difference.   python3 tools/gemm_race_check.py [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = 66 * 149
cases = [("out-proj", M, 768, 768, torch.float16), ("QKV", M, 2304, 768, torch.float16), ("FFN1", M, 3072, 768, torch.float16),
         ("FFN2", M, 768, 3072, torch.float16), ("conv (39534 x 512 x 1536)", 39534, 512, 1536, torch.float16),
         ("ragged (1000 x 640 x 832)", 1000, 640, 832, torch.float16), ("f32 19800 x 1024 x 1024", 19800, 1024, 1024, torch.float32),
         ("f32 19800 x 128 x 384", 19800, 128, 384, torch.float32), ("f32 ragged 1001*4 x 260 x 100", 4004, 260, 100, torch.float32)]
bad = 0
for name, m, n, k, dt in cases:
    g = torch.Generator().manual_seed(m + n + k)
    A = (torch.randn(m, k, generator=g) * 0.2).to(dt).to(dev)
    B = (torch.randn(n, k, generator=g) * 0.2).to(dt).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    C = torch.zeros(m, n, dtype=dt, device=dev)
    gm = ops.Gemm(m, n, k, A, B, C, lda=k, ldb=k, ldc=n, epilogue=ops.EPI_BIAS, bias=bias)
    gm()
    torch.cuda.synchronize()
    ref = C.clone()
    diff = 0
    for r in range(reps):
        C.zero_()
        gm()
        if r % 10 == 9 or r == reps - 1:
            diff += int(not torch.equal(C, ref))
    torch.cuda.synchronize()
    err = float((ref.float() - (A.float() @ B.float().t() + bias)).abs().max())
    print(f"{name:34s} {gm.kernel_name:30s} {reps} launches: {'BIT-EQUAL' if diff == 0 else f'{diff} DIFFERENT CHECKS'}   max |err| vs torch {err:.3e}", flush=True)
    bad += diff
sys.exit(1 if bad else 0)
