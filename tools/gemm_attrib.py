#!/usr/bin/env python3
"""Time attribution of the 256x256x64 phased GEMM kernel on the products it runs in the training step (fp16, real
epilogues): the SAME kernel body with parts compiled out (w2v2_tune_gemm_debug bits: 1 no DMA in the main loop, 2 no
fragment reads, 4 no MFMAs, 8 no epilogue) and the placement experiments (16 = DMA pieces between the MFMAs).

    python3 tools/gemm_attrib.py [filter]          VARIANTS=0,1,2,4,8,16 REPS=20 TRIALS=5
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
from w2v2_speaker_amd.ops import EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD, EPI_MUL, EPI_NONE

dev = "cuda"
dt = torch.float16
M = 66 * 149
H, I = 768, 3072
SHAPES = [
    ("ffn1 fwd", M, I, H, EPI_BIAS_GELU_GRAD, True),
    ("dh   bwd", M, I, H, EPI_MUL, True),
    ("plain   ", M, I, H, EPI_NONE, False),
    ("conv1   ", 66 * 4799, 512, 1536, EPI_BIAS_GELU, False),
    ("conv3   ", 66 * 1199, 512, 1536, EPI_BIAS_GELU, False),
]
NAMES = {0: "full", 1: "-dma", 2: "-reads", 4: "-mfma", 8: "-epi", 9: "-dma-epi", 11: "mfma only", 13: "reads only",
         16: "1 of 2 dma pieces in mfma", 24: "1 of 2 in mfma -epi", 32: "both dma pieces in mfma", 40: "both in mfma -epi", 48: "dma in read segment", 64: "plain stores", 128: "wt stores"}
only = sys.argv[1] if len(sys.argv) > 1 else None
variants = [int(x) for x in os.environ.get("VARIANTS", "48,32,8,40").split(",")]
reps = int(os.environ.get("REPS", "20"))
trials = int(os.environ.get("TRIALS", "5"))
_bA = torch.randn(66 * 2399, 1536, device=dev).to(dt)
_bB = torch.randn(512, 1536, device=dev).to(dt)
_bC = torch.zeros(66 * 2399, 512, dtype=dt, device=dev)
blocker = ops.Gemm(66 * 2399, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
lib = ops.lib()
print(f"# fp16, {reps} reps x {trials} trials (median), us per launch; variants: " + ", ".join(f"{v}={NAMES.get(v, v)}" for v in variants))
for name, m, n, k, epi, has_aux in SHAPES:
    if only and only not in name:
        continue
    A = torch.randn(m, k, device=dev).to(dt)
    Bw = (torch.randn(n, k, device=dev) * 0.05).to(dt)
    Cm = torch.zeros(m, n, dtype=dt, device=dev)
    aux = torch.randn(m, n, device=dev).to(dt) if has_aux else None
    bias = torch.randn(n, device=dev) if epi in (EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD) else None
    g = ops.Gemm(m, n, k, A, Bw, Cm, lda=k, ldb=k, ldc=n, epilogue=epi, bias=bias, aux=aux, ldaux=n if has_aux else 0)
    lib.w2v2_tune_gemm_kernel(4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for v in variants:
        lib.w2v2_tune_gemm_debug(v)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        ts = []
        for _ in range(trials):
            lib.w2v2_tune_gemm_debug(0)
            for _ in range(2):
                blocker()
            lib.w2v2_tune_gemm_debug(v)
            e0.record()
            for _ in range(reps):
                g()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        res[v] = sorted(ts)[len(ts) // 2]
    # the placement variants must be bit-equal to the product kernel
    lib.w2v2_tune_gemm_debug(0)
    g()
    ref = Cm.clone()
    for v in variants:
        if v in (16, 32, 48, 64, 128):
            Cm.zero_()
            lib.w2v2_tune_gemm_debug(v)
            g()
            torch.cuda.synchronize()
            assert torch.equal(Cm, ref), f"variant {v} differs from the product kernel on {name}"
    lib.w2v2_tune_gemm_debug(0)
    lib.w2v2_tune_gemm_kernel(0)
    print(f"{name} M={m:7d} N={n:5d} K={k:5d} epi={epi}  " + "  ".join(f"[{NAMES.get(v, v)}] {u:7.1f}" for v, u in res.items()),
          flush=True)
