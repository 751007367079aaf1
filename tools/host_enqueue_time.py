import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import OneCycle
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import SpeakerTrainer
dev = torch.device("cuda", 0)
cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-base")
store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994, freeze_cnn=True, embed_dim=1536)
store.init_weights(seed=20211)
plan = Plan(store, 66, 48000, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7)
tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=100), layerdrop_seed=1234, mask_seed=7)
wav, label = bench.synth_batch(66, 48000, 5994, seed=42133724, device=dev)
for _ in range(5): tr.train_step(wav, label)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); tr.train_step(wav, label); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per step: median %.2f ms min %.2f max %.2f; enqueue total %.1f ms, gpu done after %.1f ms" % (1e3*sorted(host)[10], 1e3*min(host), 1e3*max(host), 1e3*(t1-t0), 1e3*(t2-t0)))
