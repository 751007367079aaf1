import os, sys
sys.path.insert(0, "/root/repo")
import torch
from w2v2_speaker_amd import ops
dev = "cuda"
M = 9834; Mp = (M + 63) // 64 * 64
_bA = torch.randn(66 * 4799, 1536, device=dev).to(torch.bfloat16)
_bB = torch.randn(512, 1536, device=dev).to(torch.bfloat16)
_bC = torch.zeros(66 * 4799, 512, dtype=torch.bfloat16, device=dev)
blocker = ops.Gemm(66 * 4799, 512, 1536, _bA, _bB, _bC, lda=1536, ldb=1536, ldc=512)
for (no, ni) in ((1024, 512), (2048, 1024), (4096, 1024), (4096, 2048), (8192, 2048)):
    dy = torch.zeros(Mp, no, dtype=torch.bfloat16, device=dev).normal_()
    x = torch.zeros(Mp, ni, dtype=torch.bfloat16, device=dev).normal_()
    dw = torch.zeros(no, ni, device=dev)
    wg = ops.WgradGroup([(dy, x, dw, None)], M, Mp)
    for _ in range(3): wg()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): blocker()
    e0.record()
    for _ in range(10): wg()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    tiles = (no // 256) * (ni // 128)
    print(f"n_out={no} n_in={ni} tiles={tiles:4d} {us:8.1f} us  {us / 154:6.3f} us/iter/round  {wg.flops / us / 1e6:7.1f} TF")
