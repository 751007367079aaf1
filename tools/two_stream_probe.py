#!/usr/bin/env python3
"""Does this stack overlap the head / tail of one kernel with the body of another when a step is cut into two
half-batches on two HIP streams?  (VERDICT r4 item 1b asks for the per-launch fixed cost of the GEMM chain; this is
the cheapest way to find out whether stream-level concurrency can hide it before the engine is restructured for it.)

  A  one plan, B utterances, forward + head + backward on one stream                         (today's step, no Adam)
  B  two plans of B/2 utterances over the SAME store, launched alternately kernel-group by kernel-group on two
     streams (forward A, forward B, head A, head B, backward A, backward B) -- the weight gradients of the two halves
     land in the same arena slots (the second overwrites the first: the probe measures time, not values)
  C  the two half plans back to back on ONE stream (what the halving alone costs: smaller launches)

Prints ms per joint batch for each.  Run on the GPU box:  python tools/two_stream_probe.py [B] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

from bench import synth_batch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.params import ParamStore


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 66
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-base")
    store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994, freeze_cnn=True)
    store.init_weights(seed=20211)
    reg = Wav2Vec2RegularisationConfig()
    full = Plan(store, B, 48000, train=True, reg=reg, seed=7)
    h = B // 2
    halves = [Plan(store, h, 48000, train=True, reg=reg, seed=7 + i) for i in range(2)]
    wav, label = synth_batch(B, 48000, 5994, seed=1, device=dev)
    wavs, labels = [wav[:h].contiguous(), wav[h:2 * h].contiguous()], [label[:h].contiguous(), label[h:2 * h].contiguous()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def step_full(i):
        store.zero_grad(())
        full.embed(wav, None, (), i)
        full.head_forward_backward(label)
        full.backward()

    def step_two_streams(i):
        store.zero_grad(())
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                halves[k].embed(wavs[k], None, (), i)
                halves[k].head_forward_backward(labels[k])
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                halves[k].backward()
        for s in streams:
            cur.wait_stream(s)

    def step_two_serial(i):
        store.zero_grad(())
        for k in range(2):
            halves[k].embed(wavs[k], None, (), i)
            halves[k].head_forward_backward(labels[k])
            halves[k].backward()

    def fine_interleave(i):
        """Finer interleave: layer-sized host bursts alternate between the streams (forward only differs in issue order
        from step_two_streams; the hardware queues decide the rest)."""
        store.zero_grad(())
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        import threading
        def run(k):
            with torch.cuda.stream(streams[k]):
                halves[k].embed(wavs[k], None, (), i)
                halves[k].head_forward_backward(labels[k])
                halves[k].backward()
        ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for s in streams:
            cur.wait_stream(s)

    for name, fn in (("A one plan, one stream", step_full), ("B two half plans, two streams", step_two_streams),
                     ("C two half plans, one stream", step_two_serial), ("D two half plans, two streams, two host threads", fine_interleave),
                     ("A again", step_full), ("B again", step_two_streams)):
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{name:52s} {1e3 * dt:8.3f} ms per {2 * h if 'half' in name else B} utterances", flush=True)


if __name__ == "__main__":
    main()
