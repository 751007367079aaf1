#!/usr/bin/env python3
"""Split-K pairs on the phased kernel (family 6) against the 256x128 ring kernel (family 2) and an f32 product: same-XCD
exchange and the forced cross-XCD (fence) path, repeated launches (sequence numbers), several rounds (P > 128)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops as o

DEV = "cuda"
lp = torch.float16
bad = 0
for (M, N, K) in [(9834, 768, 3072), (9834, 768, 2304), (149, 768, 3072), (40000, 1024, 512), (2000, 256, 576)]:
    for epi in ("none", "bias", "add", "mul"):
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(lp).to(DEV)
        Bm = (torch.randn(N, K, generator=g) / K ** 0.5).to(lp).to(DEV)
        ref = A.float() @ Bm.float().t()
        kw, aux_in = {}, None
        if epi == "bias":
            bias = torch.randn(N, generator=g).to(DEV)
            kw.update(epilogue=o.EPI_BIAS, bias=bias)
            ref = ref + bias
        elif epi in ("add", "mul"):
            aux_in = torch.randn(M, N, generator=g).to(lp).to(DEV)
            kw.update(epilogue=o.EPI_ADD if epi == "add" else o.EPI_MUL, aux=aux_in, ldaux=N)
            ref = ref + aux_in.float() if epi == "add" else ref * aux_in.float()
        outs = {}
        for name, fam, cross in (("ring", 2, 0), ("pairs", 6, 0), ("pairs-again", 6, 0), ("cross", 6, 1), ("cross-again", 6, 1), ("pairs-after", 6, 0)):
            C = torch.full((M, N), float("nan"), dtype=lp, device=DEV)
            o.lib().w2v2_tune_gemm_kernel(fam)
            o.lib().w2v2_tune_gemm_ks_cross(cross)
            o.gemm(M, N, K, A, Bm, C, lda=K, ldb=K, ldc=N, **kw)
            torch.cuda.synchronize()
            outs[name] = C
        o.lib().w2v2_tune_gemm_kernel(0)
        o.lib().w2v2_tune_gemm_ks_cross(0)
        e_ring = float((outs["ring"].float() - ref).norm() / ref.norm())
        e_pair = float((outs["pairs"].float() - ref).norm() / ref.norm())
        same = all(torch.equal(outs["pairs"], outs[k]) for k in ("pairs-again", "cross", "cross-again", "pairs-after"))
        nan = int(torch.isnan(outs["pairs"]).sum())
        ok = same and nan == 0 and e_pair < 1.2 * e_ring + 1e-5
        print(f"M={M} N={N} K={K} {epi:5s} err ring {e_ring:.3e} pairs {e_pair:.3e} all-pair-variants-bit-equal={same} nan={nan} {'ok' if ok else 'BAD'}",
              flush=True)
        bad += not ok
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
