#!/usr/bin/env python3
"""Attention kernels alone at the step's geometry (B = 66, T = 149, 12 heads, fp16, dropout 0.1), for counter passes:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d DIR -- python3 tools/attn_pmc.py
tools/attn_pmc_report.py DIR... folds the counter_collection CSVs per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops as o

B, T, heads, d, H = 66, 149, 12, 64, 768
dev = "cuda"
qkv = torch.randn(B, T, 3 * H, device=dev).half()
dctx = torch.randn(B, T, H, device=dev).half()
ctx = torch.empty(B, T, H, dtype=torch.float16, device=dev)
lse = torch.empty(B * heads * T, device=dev)
dq = torch.zeros(B, T, 3 * H, dtype=torch.float16, device=dev)
delta = torch.empty(B * heads * T, device=dev)
for _ in range(int(os.environ.get("REPS", "6"))):
    o.attention_fwd(qkv, ctx, lse, B, T, heads, d, 0.125, 0.1, 77)
    o.attention_bwd(qkv, ctx, dctx, lse, dq, delta, B, T, heads, d, 0.125, 0.1, 77)
torch.cuda.synchronize()
