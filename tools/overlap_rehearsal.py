#!/usr/bin/env python3
"""Single-GPU rehearsal of the gradient all-reduce overlap: the training step of bench.py with a
STAND-IN for the RCCL channels on the side stream -- at every `bucket_ready` the reducer launches w2v2_traffic_probe over
the bucket's slice of the gradient arena: `channels` workgroups with an RCCL channel's LDS footprint, paced to the
bandwidth an 8-GPU xGMI ring would give the bucket, so they sit on their CUs as long as the real collective would.
What is measured is the compute-side cost (persistent GEMM grids own every CU; W2V2_RESERVE_CUS keeps some out) -- the
first setting to try on an 8-GPU lease, not a scaling number.

    python3 tools/overlap_rehearsal.py                 (sweep: spawns one child per W2V2_RESERVE_CUS value)
    python3 tools/overlap_rehearsal.py --child         (one process: baseline + every channel setting)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class ProbeReducer:
    """bucket_ready / wait / world like trainer.BucketAllReducer; the 'collective' is the traffic probe."""

    def __init__(self, store, channels, lds_bytes, gbps, bucket_merge=2):
        import torch
        from w2v2_speaker_amd import _lib
        from w2v2_speaker_amd.trainer import BucketAllReducer
        self.torch, self.lib, self.store, self.world = torch, _lib.load(), store, 1     # world 1: no 1/world scaling
        self.ranges, self.members = BucketAllReducer.merge_buckets(store.grad_buckets(), bucket_merge)
        self.channels, self.lds, self.gbps = channels, lds_bytes, gbps
        self.comm_stream = torch.cuda.Stream()
        self._issued = False
        self.bytes = 0

    def bucket_ready(self, name):
        if name not in self.ranges:
            return
        s, e = self.ranges[name]
        s = (s + 3) // 4 * 4
        if e - s < 4:
            return
        torch = self.torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        g = self.store.grad
        rc = self.lib.w2v2_traffic_probe(g.data_ptr() + 4 * s, (e - s) // 4 * 4, self.channels, self.lds, self.gbps,
                                         self.comm_stream.cuda_stream)
        assert rc == 0
        self.bytes += 4 * (e - s)
        self._issued = True

    def wait(self):
        if self._issued:
            self.torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._issued = False


def child(args):
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
    store = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=5994)
    store.init_weights(seed=20211)
    plan = Plan(store, 66, 48000, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7)
    wav, label = synth_batch(66, 48000, 5994, seed=42133724, device=dev)
    reserve = os.environ.get("W2V2_RESERVE_CUS", "0")

    def run(reducer, steps=args.steps):
        tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=10_000), layerdrop_seed=1234, mask_seed=7,
                            reducer=reducer)
        for _ in range(3):
            tr.train_step(wav, label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step(wav, label)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    base = run(None)
    print(f"reserve_cus={reserve:>3s}  no collective                                   {base:7.3f} ms/step", flush=True)
    for ch in args.channels:
        for lds in args.lds_kib:
            for gbps in args.gbps:
                red = ProbeReducer(store, ch, lds * 1024, gbps)
                ms = run(red)
                per_step = red.bytes / (args.steps + 3) / 1e6
                print(f"reserve_cus={reserve:>3s}  channels={ch:3d} lds={lds:3d} KiB pace={gbps:5.0f} GB/s ({per_step:5.0f} MB/step)  "
                      f"{ms:7.3f} ms/step  {100.0 * (ms / base - 1.0):+5.1f} %", flush=True)
    base2 = run(None)
    print(f"reserve_cus={reserve:>3s}  no collective (again)                           {base2:7.3f} ms/step", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--reserve", type=int, nargs="*", default=[0, 8, 16, 32])
    ap.add_argument("--channels", type=int, nargs="*", default=[8, 16, 32])
    ap.add_argument("--lds-kib", type=int, nargs="*", default=[64])
    ap.add_argument("--gbps", type=float, nargs="*", default=[200.0, 0.0],
                    help="bucket bytes per second of the stand-in (8-GPU ring all-reduce of 376 MB at ~350 GB/s bus bandwidth "
                         "moves a bucket at ~200 GB/s; 0 = unpaced, as fast as the channels can stream)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    print("# tools/overlap_rehearsal.py: bench.py's step (w2v2-base, B = 66, fp16) with a side-stream stand-in for the RCCL channels;"
          " step time vs the same process without it")
    for r in args.reserve:
        env = dict(os.environ, W2V2_RESERVE_CUS=str(r))
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps), "--channels",
               *map(str, args.channels), "--lds-kib", *map(str, args.lds_kib), "--gbps", *map(str, args.gbps)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        sys.stdout.write("".join(l for l in out.stdout.splitlines(True) if l.startswith("reserve_cus")))
        if out.returncode != 0:
            sys.stdout.write(out.stderr[-2000:])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
