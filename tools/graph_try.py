import sys, time; sys.path.insert(0, "/root/repo")
import torch
from w2v2_speaker_amd.config import W2V2Config
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.params import ParamStore
dev = "cuda"
cfg = W2V2Config()
st = ParamStore(cfg, dev, torch.bfloat16, head=None)
st.init_weights(1)
B, N = 66, 48000
plan = Plan(st, B, N, train=False)
wav = torch.randn(B, N, device=dev)
for _ in range(3): plan.embed(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): plan.embed(wav)
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
ref = plan.emb.clone()
static = wav.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.embed(static)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan.embed(static)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 20
print(f"eval embed B=66: eager {eager*1e3:.3f} ms, hipGraph replay {graph*1e3:.3f} ms, same result {torch.equal(ref, plan.emb)}")
