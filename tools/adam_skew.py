#!/usr/bin/env python3
"""Does the fused Adam's bandwidth depend on how its five streams (p, g, m, v f32 + the 16-bit copy) are placed relative to
each other?  All carved from ONE allocation at a byte skew between consecutive arrays (0 = what separate 2 MiB-aligned
allocations give).  Box-to-box the in-step kernel measured 485-613 us this round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
n = 99_400_000 // 64 * 64
dev = "cuda"
big = torch.zeros(5 * (n + (1 << 22)), dtype=torch.float32, device=dev)
sc = torch.tensor([16384.0, 0, 0, 0, 0, 0, 0, 0], device=dev)
def carve(skew_bytes):
    stride = ((n * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20)          # 2 MiB-aligned slots
    out = []
    for i in range(5):
        off = (i * stride + i * skew_bytes) // 4
        out.append(big[off:off + n])
    return out
for skew in (0, 256, 1024, 4096, 4096 + 256, 65536 + 4096, (1 << 20) + 4096 + 256):
    p, g, m, v, pbf = carve(skew)
    pb = pbf.view(torch.float16)[:n]
    g.normal_(); g.mul_(1e-3); p.normal_(); m.zero_(); v.zero_()
    for _ in range(3):
        ops.adam_step(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 10, 1.0, sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            ops.adam_step(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 10, 1.0, sc)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    us = sorted(ts)[2]
    print(f"skew {skew:8d} B: {us:7.1f} us  {30.0 * n / us / 1e6:5.2f} TB/s", flush=True)
