"""Does a GEMM slow down when its output (and input) buffers are cold, as inside the training step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
dev = "cuda"
M, N, K = 66 * 149, 2304, 768
NB = int(os.environ.get("NB", "24"))
W = torch.randn(N, K, device=dev).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NB)]
Cs = [torch.zeros(M, N, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
for mode in ("hot A, hot C", "hot A, cold C", "cold A, cold C"):
    gs = []
    for i in range(NB):
        a = As[0] if "hot A" in mode else As[i]
        c = Cs[0] if "hot C" in mode else Cs[i]
        gs.append(ops.Gemm(M, N, K, a, W, c, lda=K, ldb=K, ldc=N, epilogue=ops.EPI_BIAS, bias=bias))
    for g in gs:
        g()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        for g in gs:
            g()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * NB)
    print(f"{mode:16s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
