"""Fixed (prologue + epilogue + launch) vs per-K-tile cost of the GEMM kernels: sweep K at the encoder's M, N."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"
M = 66 * 149
_bA = torch.randn(66 * 4799, 1536, device=dev).to(torch.bfloat16)
_bB = torch.randn(512, 1536, device=dev).to(torch.bfloat16)
_bC = torch.zeros(66 * 4799, 512, dtype=torch.bfloat16, device=dev)
blocker = ops.Gemm(66 * 4799, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
for n in (768, 2304, 3072):
    for k in (64, 128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, k, device=dev).to(torch.bfloat16)
        B = torch.randn(n, k, device=dev).to(torch.bfloat16)
        C = torch.zeros(M, n, dtype=torch.bfloat16, device=dev)
        g = ops.Gemm(M, n, k, A, B, C, lda=k, ldb=k, ldc=n)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            blocker()
        e0.record()
        for _ in range(20):
            g()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"N={n:5d} K={k:5d} ktiles={k // 64:3d}  {us:8.1f} us  {2.0 * M * n * k / us / 1e6:8.1f} TF  [{g.kernel_name}]")
