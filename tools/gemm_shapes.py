#!/usr/bin/env python3
"""Stand-alone timing of the encoder / conv products of the B=66 training step WITH their real epilogues
(bias, bias+GELU dual store, GELU' with aux read, residual add, two-term weight columns), fp16 operands.
tools/gemm_bench.py times the plain product; the in-step launches carry 60-120 MB of epilogue traffic that it
does not show.  Kernel choice follows the library's dispatch; the A/B environment switches
of csrc/gemm.hip (W2V2_NO_GEMM_PH, W2V2_NO_GLDS3, W2V2_NO_GLDS) select alternatives per process.

    python3 tools/gemm_shapes.py [filter] [--hipblaslt]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
from w2v2_speaker_amd.ops import EPI_ADD, EPI_BIAS, EPI_BIAS_GELU, EPI_GELU_BWD, EPI_NONE

dev = "cuda"
dt = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float16
M = 66 * 149
H, I = 768, 3072
SHAPES = [
    # name, M, N, K, epilogue, aux?, two-term columns from (or None)
    ("qkv  fwd", M, 3 * H, H, EPI_BIAS, False, 2 * H),
    ("out  fwd", M, H, H, EPI_BIAS, False, 0),
    ("ffn1 fwd", M, I, H, EPI_BIAS_GELU, True, None),
    ("ffn2 fwd", M, H, I, EPI_BIAS, False, None),
    ("dh   bwd", M, I, H, EPI_GELU_BWD, True, None),
    ("dx1  bwd", M, H, I, EPI_ADD, True, None),
    ("dctx bwd", M, H, H, EPI_NONE, False, None),
    ("dx   bwd", M, H, 3 * H, EPI_ADD, True, None),
    ("conv1   ", 66 * 4799, 512, 1536, EPI_BIAS_GELU, False, None),
    ("conv2   ", 66 * 2399, 512, 1536, EPI_BIAS_GELU, False, None),
    ("conv3   ", 66 * 1199, 512, 1536, EPI_BIAS_GELU, False, None),
    ("conv4   ", 66 * 599, 512, 1536, EPI_BIAS_GELU, False, None),
    ("conv5   ", 66 * 299, 512, 1024, EPI_BIAS_GELU, False, None),
    ("conv6   ", 66 * 149, 512, 1024, EPI_BIAS_GELU, False, None),
    ("proj    ", M, H, 512, EPI_BIAS, False, None),
]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
only = args[0] if args else None
FAM = {0: "auto", 1: "128x128", 2: "256x128 ring", 4: "256x256x64 phased"}
fams = [int(x) for x in os.environ.get("FAMILIES", "0").split(",")]
vs_lib = "--hipblaslt" in sys.argv

_bA = torch.randn(66 * 2399, 1536, device=dev).to(dt)
_bB = torch.randn(512, 1536, device=dev).to(dt)
_bC = torch.zeros(66 * 2399, 512, dtype=dt, device=dev)
blocker = ops.Gemm(66 * 2399, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
reps = int(os.environ.get("REPS", "20"))
trials = int(os.environ.get("TRIALS", "5"))
print(f"# dtype {dt}, {reps} reps x {trials} trials (median), env: " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("W2V2_")))
for name, m, n, k, epi, has_aux, two in SHAPES:
    if only and only not in name:
        continue
    A = torch.randn(m, k, device=dev).to(dt)
    Bw = (torch.randn(2, n, k, device=dev) * 0.05).to(dt)        # plane 0 = hi, plane 1 = lo (two-term)
    Cm = torch.zeros(m, n, dtype=dt, device=dev)
    aux = (torch.randn(m, n, device=dev)).to(dt) if has_aux else None
    bias = torch.randn(n, device=dev) if epi in (EPI_BIAS, EPI_BIAS_GELU) else None
    g = ops.Gemm(m, n, k, A, Bw[0], Cm, lda=k, ldb=k, ldc=n, epilogue=epi, bias=bias, aux=aux, ldaux=n if has_aux else 0,
                 b_lo=Bw[1] if two is not None else None, n_ext_from=two or 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per_fam = {}
    for fam in fams:
        if fam and two is not None:
            continue
        ops.lib().w2v2_tune_gemm_kernel(0)
        for _ in range(3):
            blocker()
        ops.lib().w2v2_tune_gemm_kernel(fam)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        ts = []
        for _ in range(trials):
            ops.lib().w2v2_tune_gemm_kernel(0)
            for _ in range(3):
                blocker()
            ops.lib().w2v2_tune_gemm_kernel(fam)
            e0.record()
            for _ in range(reps):
                g()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        per_fam[fam] = sorted(ts)[len(ts) // 2]
    ops.lib().w2v2_tune_gemm_kernel(0)
    if len(fams) > 1:
        print(f"{name} M={m:7d} N={n:5d} K={k:5d} epi={epi}  " + "  ".join(f"[{FAM[f]}] {u:7.1f} us" for f, u in per_fam.items()),
              flush=True)
        continue
    us = per_fam[fams[0]]
    ext = 2.0 * m * (n - two) * k if two is not None else 0.0
    lib = ""
    if vs_lib and two is None:
        out = torch.empty(m, n, dtype=dt, device=dev)
        Bt = Bw[0].t()
        for _ in range(3):
            torch.matmul(A, Bt, out=out)
        for _ in range(3):
            blocker()
        e0.record()
        for _ in range(reps):
            torch.matmul(A, Bt, out=out)
        e1.record()
        torch.cuda.synchronize()
        lus = e0.elapsed_time(e1) * 1e3 / reps
        lib = f"   hipBLASLt (plain product) {lus:7.1f} us"
    print(f"{name} M={m:7d} N={n:5d} K={k:5d} epi={epi} {g.kernel_name:24s} {us:8.1f} us  "
          f"{2.0 * m * n * k / us / 1e6:7.1f} TF/s alg  {(2.0 * m * n * k + ext) / us / 1e6:7.1f} TF/s exec  "
          f"{g.bytes / us / 1e3:6.2f} TB/s alg{lib}", flush=True)
