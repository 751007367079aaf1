"""Micro-benchmark of the GEMM kernel on the shapes of the w2v2-base training step (B=66, T=149)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops

dev = "cuda"
M = 66 * 149
shapes = [
    # name, M, N, K, transA, transB, out dtype, split
    ("fwd qkv   NT", M, 2304, 768, False, False, torch.bfloat16, 1),
    ("fwd out   NT", M, 768, 768, False, False, torch.bfloat16, 1),
    ("fwd ffn1  NT", M, 3072, 768, False, False, torch.bfloat16, 1),
    ("fwd ffn2  NT", M, 768, 3072, False, False, torch.bfloat16, 1),
    ("conv1     NT", 66 * 4799, 512, 1536, False, False, torch.bfloat16, 1),
    ("conv4     NT", 66 * 599, 512, 1536, False, False, torch.bfloat16, 1),
    ("dx  ffn2  NN", M, 3072, 768, False, True, torch.bfloat16, 1),
    ("dx  ffn1  NN", M, 768, 3072, False, True, torch.bfloat16, 1),
    ("dW  ffn1  TT", 3072, 768, M, True, True, torch.float32, 3),
    ("dW  ffn2  TT", 768, 3072, M, True, True, torch.float32, 3),
    ("dW  qkv   TT", 2304, 768, M, True, True, torch.float32, 4),
    ("dW  out   TT", 768, 768, M, True, True, torch.float32, 14),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
# a ~3 ms blocker so the host enqueues the timed launches while the GPU is still busy: the events then
# bracket back-to-back device execution, not host launch latency
_bA = torch.randn(66 * 4799, 1536, device=dev).to(torch.bfloat16)
_bB = torch.randn(512, 1536, device=dev).to(torch.bfloat16)
_bC = torch.zeros(66 * 4799, 512, dtype=torch.bfloat16, device=dev)
blocker = ops.Gemm(66 * 4799, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
for name, m, n, k, ta, tb, cdt, split in shapes:
    if only and only not in name:
        continue
    A = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
    B = torch.randn((k, n) if tb else (n, k), device=dev).to(torch.bfloat16)
    C = torch.zeros(m, n, dtype=cdt, device=dev)
    g = ops.Gemm(m, n, k, A, B, C, lda=m if ta else k, ldb=n if tb else k, ldc=n, transA=ta, transB=tb,
                 split_k=split, accumulate=split > 1)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get('REPS', '20'))
    trials = []
    for _ in range(int(os.environ.get('TRIALS', '5'))):
        for _ in range(3):
            blocker()
        e0.record()
        for _ in range(reps):
            g()
        e1.record()
        torch.cuda.synchronize()
        trials.append(e0.elapsed_time(e1) * 1e3 / reps)
    us = sorted(trials)[len(trials) // 2]          # median of the batches (boxes jitter by several %)
    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
    got = C.float() / (reps * len(trials) + 3 if split > 1 else 1)
    err = float((got - ref).norm() / ref.norm())
    lib = ""
    if os.environ.get("VS_HIPBLASLT") and cdt == torch.bfloat16:
        # calibration only: the vendor library on the same operands (torch.matmul -> hipBLASLt)
        Am = A.t() if ta else A
        Bm = B if tb else B.t()
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            torch.matmul(Am, Bm, out=out)
        for _ in range(3):
            blocker()
        e0.record()
        for _ in range(reps):
            torch.matmul(Am, Bm, out=out)
        e1.record()
        torch.cuda.synchronize()
        lus = e0.elapsed_time(e1) * 1e3 / reps
        lib = f"   hipBLASLt {lus:8.1f} us {2.0 * m * n * k / lus / 1e6:7.1f} TFLOP/s"
    print(f"{name}  M={m:7d} N={n:5d} K={k:6d}  {us:9.1f} us  {2.0 * m * n * k / us / 1e6:8.1f} TFLOP/s  rel-err {err:.1e}{lib}")

# grouped weight-gradient launch of one transformer block
if not only or "wgrad" in only:
    Mp = (M + 63) // 64 * 64
    H, I = 768, 3072
    mk = lambda c: torch.zeros(Mp, c, dtype=torch.bfloat16, device=dev).normal_()
    probs = [(mk(H), mk(I), torch.zeros(H, I, device=dev), torch.zeros(H, device=dev)),
             (mk(I), mk(H), torch.zeros(I, H, device=dev), torch.zeros(I, device=dev)),
             (mk(H), mk(H), torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)),
             (mk(3 * H), mk(H), torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev))]
    if os.environ.get('W2V2_NO_DBIAS'):
        probs = [(a, b, c, None) for a, b, c, _ in probs]
    wg = ops.WgradGroup(probs, M, Mp)
    for _ in range(3):
        wg()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        blocker()
    e0.record()
    for _ in range(10):
        wg()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    print(f"wgrad grouped (dW2,dW1,dWo,dWqkv + biases)  {us:9.1f} us  {wg.flops / us / 1e6:8.1f} TFLOP/s")

# two transformer blocks' weight gradients in one launch (8 problems -> 216 tiles of 256x256, wgrad_grouped_ring4)
if not only or "wgrad" in only:
    probs2 = probs + [(mk(H), mk(I), torch.zeros(H, I, device=dev), torch.zeros(H, device=dev)),
                      (mk(I), mk(H), torch.zeros(I, H, device=dev), torch.zeros(I, device=dev)),
                      (mk(H), mk(H), torch.zeros(H, H, device=dev), torch.zeros(H, device=dev)),
                      (mk(3 * H), mk(H), torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev))]
    if os.environ.get('W2V2_NO_DBIAS'):
        probs2 = [(a, b, c, None) for a, b, c, _ in probs2]
    wg2 = ops.WgradGroup(probs2, M, Mp)
    for _ in range(3):
        wg2()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        blocker()
    e0.record()
    for _ in range(10):
        wg2()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    print(f"wgrad grouped, two blocks in one launch        {us:9.1f} us  {wg2.flops / us / 1e6:8.1f} TFLOP/s")

