#!/usr/bin/env python3
"""Where the time of a split-K pair launch goes: per-workgroup s_memrealtime stamps of the stamping variant of the phased
split-K kernel (w2v2_tune_gemm_ks_stamps), fp16, the FFN2-shaped products of the B = 66 step.

    python tools/ksplit_stamps.py > profiles/r05_ksplit_stamps.txt
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from w2v2_speaker_amd import ops as o

DEV = "cuda"
lib = o.lib()
M = 66 * 149
for name, N, K, epi in (("ffn2 fwd", 768, 3072, "bias"), ("dx1 bwd", 768, 3072, "add"), ("dx bwd", 768, 2304, "add")):
    A = torch.randn(M, K, device=DEV).half()
    Bm = (torch.randn(N, K, device=DEV) * 0.05).half()
    C = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    kw = dict(epilogue=o.EPI_BIAS, bias=torch.randn(N, device=DEV)) if epi == "bias" else \
        dict(epilogue=o.EPI_ADD, aux=torch.randn(M, N, device=DEV).half(), ldaux=N)
    big = torch.randn(64 << 20, device=DEV)
    lib.w2v2_tune_gemm_kernel(6)
    lib.w2v2_tune_gemm_ks_stamps(1)
    rows = []
    for rep in range(12):
        big.mul_(1.0001)                                   # (something else between the launches, as in the step)
        o.gemm(M, N, K, A, Bm, C, lda=K, ldb=K, ldc=N, **kw)
        out = np.zeros((256, 6), dtype=np.uint64)
        assert lib.w2v2_tune_gemm_ks_read_stamps(ctypes.c_void_p(o.stream()), out.ctypes.data_as(ctypes.c_void_p)) == 0
        if rep < 2:
            continue
        t = out[out[:, 0] > 0].astype(np.float64) * 0.01    # us
        t0 = t[:, 0].min()
        rows.append([np.median(t[:, 0] - t0), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]),
                     np.median(t[:, 4] - t[:, 3]), np.median(t[:, 5] - t[:, 4]), (t[:, 5].max() - t0), (t[:, 1] - t0).max(),
                     (t[:, 3] - t[:, 2]).max(), len(t)])
    lib.w2v2_tune_gemm_kernel(0)
    lib.w2v2_tune_gemm_ks_stamps(0)
    r = np.median(np.array(rows), axis=0)
    print(f"{name} M={M} N={N} K={K}: {int(r[9])} workgroups; medians over workgroups (us): start skew {r[0]:.2f} | prologue + main loop "
          f"{r[1]:.2f} | swap + publish {r[2]:.2f} | flag wait {r[3]:.2f} | read + add {r[4]:.2f} | epilogue + drain {r[5]:.2f} || "
          f"first start -> last end {r[6]:.2f}, slowest main loop ends at {r[7]:.2f}, longest flag wait {r[8]:.2f}")
