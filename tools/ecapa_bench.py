"""ECAPA-TDNN training step (BASELINE configs[4]: fbank-like [B, 300, 40] input -> 192-d embedding -> AAM(5994)) on
one MI355X: utterances/sec and the kernel mix.  Single GPU only; the headline metric stays bench.py (wav2vec2)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd.ecapa import EcapaConfig, EcapaPlan, EcapaStore, EcapaTrainer
from w2v2_speaker_amd.optim.schedule import OneCycle

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=66)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
a = ap.parse_args()
dev = "cuda"
cfg = EcapaConfig()
st = EcapaStore(cfg, dev, torch.bfloat16 if a.dtype == "bf16" else torch.float32, num_speakers=5994)
st.init_weights(1)
plan = EcapaPlan(st, a.batch, a.frames, train=True)
tr = EcapaTrainer(st, plan, OneCycle(max_lr=1e-3, total_steps=max(a.steps + a.warmup + 1, 10)))
g = torch.Generator().manual_seed(0)
feat = torch.randn(a.batch, a.frames, cfg.input_mel_coefficients, generator=g).to(dev)
label = torch.randint(0, 5994, (a.batch,), generator=g).to(dev)
for _ in range(a.warmup):
    tr.train_step(feat, label)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss, _ = tr.train_step(feat, label)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# algorithmic FLOPs per utterance (2*MAC), forward: convs + ASP + fc + head; training ~ 3x forward
T, C = a.frames, cfg.channels
w = C[1] // cfg.res2net_scale
fwd = 2 * T * (cfg.input_mel_coefficients * cfg.kernel_sizes[0] * C[0]
               + 3 * (2 * C[1] * C[1] + (cfg.res2net_scale - 1) * w * w * 3) + C[-1] * C[-1]
               + C[-1] * cfg.attention_channels * 2) + 2 * (2 * C[-1] * cfg.lin_neurons + cfg.lin_neurons * 5994)
print(json.dumps({"metric": "utterances/sec (ECAPA-TDNN C=1024 + AAM-softmax training step, 300 fbank frames)",
                  "value": round(a.batch * a.steps / dt, 1), "ms_per_step": round(1e3 * dt / a.steps, 3),
                  "batch": a.batch, "dtype": a.dtype, "fwd_gflop_per_utt": round(fwd / 1e9, 3),
                  "model_tflops": round(3 * fwd * a.batch * a.steps / dt / 1e12, 1), "final_loss": round(float(loss), 4)}))
