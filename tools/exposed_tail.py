#!/usr/bin/env python3
"""What the gradient all-reduce can NOT hide, from measured single-GPU step timing.

Runs bench.py's training step on one GPU with a reducer that only RECORDS a HIP event at every `bucket_ready`
(trainer.BucketAllReducer's hook; no collective), plus events at step start, end of backward and end of Adam; averaged
over steps without LayerDrop skips.  From the measured ready times and the bucket sizes it replays the side stream of
the data-parallel job for an 8-GPU ring:

    start_i = max(ready_i, end_{i-1});  end_i = start_i + LAT + bytes_i / algbw        (buckets in firing order)
    exposed = max(0, end_last - backward_end)          step_ddp = step_1gpu + exposed (+ the overlap cost, r04 rehearsal)

for RCCL all-reduce algorithm bandwidths of 150 / 250 / 350 GB/s (algbw = bytes / time as nccl-tests reports it; one ring
over xGMI is per-link bound at ~153 GB/s, striping over the 7 links gives more) and bucket_merge in {1, 2, 4}.
Nothing here is a measured multi-GPU number: it bounds the scaling efficiency from the compute side.

    python tools/exposed_tail.py > profiles/r05_exposed_tail.txt          (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch

from bench import synth_batch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import OneCycle
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import BucketAllReducer, SpeakerTrainer

LAT_US = 30.0           # per-collective launch + ring start-up latency (RCCL small-message time on xGMI; assumed)


class Recorder:
    world = 1

    def __init__(self):
        self.events = []

    def bucket_ready(self, name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events.append((name, ev))

    def wait(self):
        pass


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # TAIL_MODEL=large TAIL_BATCH=32 TAIL_SAMPLES=80000: BASELINE configs[3]'s per-GPU share (W2V2_WGRAD_GROUP=2 / 4 to compare)
    model = os.environ.get("TAIL_MODEL", "base")
    nb, ns = int(os.environ.get("TAIL_BATCH", "66")), int(os.environ.get("TAIL_SAMPLES", "48000"))
    cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-" + model)
    store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994, freeze_cnn=True)
    store.init_weights(seed=20211)
    plan = Plan(store, nb, ns, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7)
    print(f"# model {model}, {nb} utterances of {ns} samples per GPU, weight-gradient groups of {getattr(plan, 'wg_group', 2)}")
    rec = Recorder()
    tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=100), reducer=rec)
    wav, label = synth_batch(nb, ns, 5994, seed=42133724, device=dev)
    for _ in range(5):
        tr.train_step(wav, label, skip_layers=())
    torch.cuda.synchronize()
    raw = store.grad_buckets()
    sizes = {n: 4 * (e - s) for n, s, e in raw}
    ready, bwd_end, step_ms = {}, [], []
    for _ in range(12):
        rec.events = []
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        # train_step = zero_grad, forward, head, backward (fires bucket_ready), wait, Adam
        plan_bw = plan.backward

        def timed_backward(*a, **k):
            plan_bw(*a, **k)
            e1.record()
        plan.backward = timed_backward
        tr.train_step(wav, label, skip_layers=())
        plan.backward = plan_bw
        e2.record()
        torch.cuda.synchronize()
        for n, ev in rec.events:
            ready.setdefault(n, []).append(e0.elapsed_time(ev))
        bwd_end.append(e0.elapsed_time(e1))
        step_ms.append(e0.elapsed_time(e2))
    ready = {n: float(np.median(v)) for n, v in ready.items()}
    t_bwd, t_step = float(np.median(bwd_end)), float(np.median(step_ms))
    print(f"# exposed tail of the gradient all-reduce: w2v2-base, B = 66 per GPU, fp16 mode, no LayerDrop skips in these steps")
    print(f"# measured on ONE GPU (HIP events, median of 12 steps): step {t_step:.3f} ms, backward ends at {t_bwd:.3f} ms, "
          f"Adam + scaler {t_step - t_bwd:.3f} ms")
    print("# bucket (raw)        MB      final at (ms after step start)   ms before the end of backward")
    for n, s, e in raw:
        print(f"  {n:12s} {sizes[n] / 1e6:9.2f} {ready[n]:12.3f} {t_bwd - ready[n]:30.3f}")
    print(f"# assumed per-collective latency {LAT_US:.0f} us; 8 GPUs; all-reduce time = latency + bytes / algbw")
    print("# merge  algbw(GB/s)  collectives  comm busy (ms)  exposed tail (ms)  step_ddp/step_1gpu  efficiency bound")
    best = None
    for merge in (1, 2, 4):
        ranges, members = BucketAllReducer.merge_buckets(raw, merge)
        order = [(n, sum(sizes[m] for m in members[n]), ready[n]) for n in ranges]       # fires when the LAST member is final
        order.sort(key=lambda t: t[2])
        for bw in (150.0, 250.0, 350.0):
            t_end, busy = 0.0, 0.0
            for n, nbytes, rdy in order:
                dur = LAT_US * 1e-3 + nbytes / (bw * 1e9) * 1e3
                t_end = max(rdy, t_end) + dur
                busy += dur
            exposed = max(0.0, t_end - t_bwd)
            eff = t_step / (t_step + exposed)
            print(f"  {merge:5d} {bw:11.0f} {len(order):12d} {busy:15.3f} {exposed:18.3f} {(t_step + exposed) / t_step:19.4f} {eff:17.4f}")
            if bw == 250.0 and (best is None or exposed <= best[1] + 1e-9):       # ties: fewer, larger collectives
                best = (merge, exposed)
    print(f"# default bucket_merge by this table (250 GB/s column): {best[0]} (exposed {best[1]:.3f} ms); the r04 rehearsal adds "
          "~4-5 % of step time for the CUs the collective's channels occupy while they overlap (profiles/r04_overlap_rehearsal.txt)")
    print("# reading: the LAST bucket (projection + masked embed + feature LayerNorm: final only when backward ends) is exposed in "
          "full -- since round 5 it is 1.6 MB; the pos-conv pair (19 MB) became its own earlier bucket.  What precedes it "
          "overlaps unless the ring is slower than the backward that remains.")


if __name__ == "__main__":
    main()
