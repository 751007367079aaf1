"""Time the positional-conv weight gradient: correlation kernel vs the implicit-GEMM path (B=66, T=149, base)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"
B, T, G, Cg, K = 66, 149, 16, 48, 128
H, M, Tp = G * Cg, 66 * 149, 149 + 127
x = torch.randn(B, T, H, device=dev).to(torch.bfloat16)
dY = torch.randn(M, H, device=dev).to(torch.bfloat16)
xg = torch.zeros(B, G, Tp, Cg, dtype=torch.bfloat16, device=dev)
ops.posconv_regroup(x, xg, B, T, H, G, K, K // 2)
dwf = torch.zeros(G, K * Cg, Cg, dtype=torch.float32, device=dev)
dwg = torch.zeros_like(dwf)
gemm = ops.Gemm(K * Cg, Cg, M, xg, dY, dwg, lda=Cg, ldb=H, ldc=Cg, transA=True, transB=True, a_seg=(T, G * Tp * Cg),
                batch=G, batch_inner=G, a_strides=(0, Tp * Cg), b_strides=(0, Cg), c_strides=(0, K * Cg * Cg))
fl = 2.0 * M * H * Cg * K


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


t1 = timeit(lambda: ops.posconv_wgrad(dY, xg, dwf, B, T, H, G, K))
t2 = timeit(gemm)
err = float((dwf - dwg).abs().max() / dwg.abs().max())
print(f"correlation kernel {t1:8.1f} us {fl / t1 / 1e6:7.1f} TFLOP/s | implicit GEMM {t2:8.1f} us {fl / t2 / 1e6:7.1f} TFLOP/s | rel diff {err:.1e}")
