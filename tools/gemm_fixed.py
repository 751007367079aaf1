"""Fixed cost of a GEMM launch: K = 64 products at several grid sizes against a plain 15 MB store and an empty launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.empty(64 << 20, device=dev)
    big.zero_()                       # ~45 us blocker so the launches queue up behind it
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M, N, K in [(256, 128, 64), (256, 768, 64), (9834, 768, 64), (9834, 768, 128), (9834, 768, 768), (9834, 128, 768),
                (4917, 768, 768), (2048, 768, 768)]:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.zeros(M, N, dtype=torch.float16, device=dev)
    g = ops.Gemm(M, N, K, A, B, C, lda=K, ldb=K, ldc=N)
    print(f"gemm M={M} N={N} K={K} [{g.kernel_name}] {timed(g):7.2f} us")
x = torch.zeros(9834 * 768, dtype=torch.float16, device=dev)
y = torch.zeros(9834 * 768, dtype=torch.float16, device=dev)
print(f"torch fill 15 MB   {timed(lambda: x.zero_()):7.2f} us")
print(f"torch copy 15 MB   {timed(lambda: y.copy_(x)):7.2f} us")
t = torch.zeros(64, dtype=torch.float16, device=dev)
print(f"torch fill 128 B   {timed(lambda: t.zero_()):7.2f} us")
f = torch.zeros(64, device=dev)
print(f"cast 64 elements   {timed(lambda: ops.cast(f, t)):7.2f} us")
