#!/usr/bin/env python3
"""Time the fused Adam launch over the w2v2-base arena (99.4 M trainable parameters, fp16 operand copy):
30 B/parameter of algorithmic HBM traffic.  Knobs: W2V2_ADAM_U, W2V2_ADAM_BLOCKS (csrc/optim.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
n = int(os.environ.get("ADAM_N", "99400000")) // 64 * 64
dev = "cuda"
p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
pb = torch.empty(n, dtype=torch.float16, device=dev)
sc = torch.tensor([16384.0, 0, 0, 0], device=dev)
for _ in range(3):
    ops.adam_step(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 10, 1.0, sc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(10):
        ops.adam_step(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 10, 1.0, sc)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
us = sorted(ts)[2]
print(f"adam U={os.environ.get('W2V2_ADAM_U', '1')} blocks<={os.environ.get('W2V2_ADAM_BLOCKS', '8192')}: {us:7.1f} us  {30.0 * n / us / 1e6:5.2f} TB/s")
