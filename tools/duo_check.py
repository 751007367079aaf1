#!/usr/bin/env python3
"""Bit-equality of the two-workgroups-per-CU kernel (family 5) against the 256x128 ring kernel (family 2), fp16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops as o

DEV = "cuda"
lp = torch.float16
bad = 0
for (M, N, K) in [(9834, 3072, 768), (9834, 768, 3072), (19734, 512, 1024), (1000, 256, 64), (300, 128, 128)]:
    for epi in ("none", "bias", "bias_gelu_grad", "mul", "add"):
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(lp).to(DEV)
        Bm = (torch.randn(N, K, generator=g) / K ** 0.5).to(lp).to(DEV)
        kw, aux_in = {}, None
        if epi in ("bias", "bias_gelu_grad"):
            kw.update(epilogue=o.EPI_BIAS if epi == "bias" else o.EPI_BIAS_GELU_GRAD, bias=torch.randn(N, generator=g).to(DEV))
        elif epi in ("add", "mul"):
            aux_in = torch.randn(M, N, generator=g).to(lp).to(DEV)
            kw.update(epilogue=o.EPI_ADD if epi == "add" else o.EPI_MUL)
        outs = {}
        for fam in (5, 2):
            C = torch.full((M, N), float("nan"), dtype=lp, device=DEV)
            k2 = dict(kw)
            if epi == "bias_gelu_grad":
                k2.update(aux=torch.full((M, N), float("nan"), dtype=lp, device=DEV), ldaux=N)
            elif aux_in is not None:
                k2.update(aux=aux_in, ldaux=N)
            o.lib().w2v2_tune_gemm_kernel(fam)
            o.gemm(M, N, K, A, Bm, C, lda=K, ldb=K, ldc=N, **k2)
            torch.cuda.synchronize()
            outs[fam] = (C, k2.get("aux") if epi == "bias_gelu_grad" else None)
        o.lib().w2v2_tune_gemm_kernel(0)
        eq = torch.equal(outs[5][0], outs[2][0]) and (outs[5][1] is None or torch.equal(outs[5][1], outs[2][1]))
        d = float((outs[5][0].float() - outs[2][0].float()).abs().max())
        print(f"M={M} N={N} K={K} {epi:15s} bit-equal={eq} maxdiff={d:.3e} nan={int(torch.isnan(outs[5][0]).sum())}", flush=True)
        bad += not eq
print("FAILED" if bad else "ALL EQUAL")
sys.exit(1 if bad else 0)
