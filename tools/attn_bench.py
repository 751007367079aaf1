"""Micro-benchmark of the fused attention kernels (B=66, T=149, 12 heads x 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops
dev = "cuda"
B, T, heads, d = 66, int(os.environ.get("T", "149")), 12, 64
H = heads * d
qkv = torch.randn(B, T, 3 * H, device=dev).to(torch.bfloat16)
dctx = torch.randn(B, T, H, device=dev).to(torch.bfloat16)
ctx = torch.zeros(B, T, H, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B * heads * T, device=dev)
dqkv = torch.zeros(B, T, 3 * H, dtype=torch.bfloat16, device=dev)
delta = torch.zeros(B * heads * T, device=dev)
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for geom in os.environ.get("GEOMS", "64,32,32nodma").split(","):
  os.environ["W2V2_ATTN_GEOM"] = geom[:2]        # read per call by attention.hip
  os.environ.pop("W2V2_ATTN_KV_NO_DMA", None)
  if geom.endswith("nodma"): os.environ["W2V2_ATTN_KV_NO_DMA"] = "1"
  for p in (0.0, 0.1):
    f = timeit(lambda: ops.attention_fwd(qkv, ctx, lse, B, T, heads, d, d ** -0.5, p, 1))
    b = timeit(lambda: ops.attention_bwd(qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, d, d ** -0.5, p, 1))
    fl = 4.0 * B * heads * T * T * d
    print(f"geom {geom} T={T} p={p}: fwd {f:7.1f} us ({fl / f / 1e6:6.1f} TF)   bwd {b:7.1f} us ({2.5 * fl / b / 1e6:6.1f} TF)")
