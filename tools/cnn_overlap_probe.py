#!/usr/bin/env python3
"""Can the frozen CNN forward of step n + 1 hide under the backward of step n?  (The conv feature extractor does not
depend on anything the optimiser writes: a side stream can compute the next batch's features while the main stream runs
the encoder backward, whose single-round launches leave 22-40 of the 256 CUs idle.)

  A  train step as today (CNN + encoder forward, head, backward, Adam) on one stream
  B  the same step + a SECOND plan's conv_features() on a side stream, enqueued when the main stream starts its backward
  C  the conv features alone
If B - A is much less than C the overlap works; the pipelined step would then cost about A - C + (B - A)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from bench import synth_batch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import OneCycle
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import SpeakerTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-base")
store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994, freeze_cnn=True)
store.init_weights(seed=20211)
reg = Wav2Vec2RegularisationConfig()
plan = Plan(store, 66, 48000, train=True, reg=reg, seed=7)
side_plan = Plan(store, 66, 48000, train=False)              # only its conv buffers / descriptors are used
tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=1000))
wav, label = synth_batch(66, 48000, 5994, seed=1, device=dev)
side = torch.cuda.Stream()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def step(overlap):
    store.zero_grad(())
    plan.embed(wav, None, (), tr.step)
    plan.head_forward_backward(label)
    if overlap:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side_plan.conv_features(wav)
    plan.backward()
    lr, b1 = tr.schedule.at(tr.step)
    store.adam_step(lr, b1, tr.beta2, tr.eps)
    if overlap:
        torch.cuda.current_stream().wait_stream(side)
    tr.step += 1

def timeit(fn, n):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

for rnd in range(2):
    a = timeit(lambda: step(False), steps)
    b = timeit(lambda: step(True), steps)
    c = timeit(lambda: side_plan.conv_features(wav), steps)
    print(f"A step {a:.3f} ms   B step + side-stream CNN {b:.3f} ms   C CNN alone {c:.3f} ms   ->  B - A = {b - a:.3f} ms, "
          f"pipelined step ~ {b - c:.3f} ms ({100 * (b - c - a) / a:+.1f} %)", flush=True)
