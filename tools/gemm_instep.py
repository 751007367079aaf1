#!/usr/bin/env python3
"""Per-shape IN-STEP table of the matrix-core launches of one training step.

Two phases, both on the GPU box:

  run   : python3 tools/gemm_instep.py run  <seq.json> [--steps K]      (put this one behind `rocprofv3 --kernel-trace -d DIR --`)
          runs W warm-up + K timed training steps of the bench workload (B=66, 3 s, fp16, regularisation on) and
          records, in launch order, the shape key of every GEMM / grouped weight-gradient launch of the K steps.
  join  : python3 tools/gemm_instep.py join <seq.json> <results.db> [out.txt]
          reads the kernel trace (rocpd sqlite), keeps the dispatches whose kernel name matches the launch log, checks
          that the two sequences agree one to one (kernel class per launch) and prints avg / min duration, TFLOP/s and
          ms/step per (M, N, K, epilogue, kernel).

The join is by ORDER: the stream is in-order and the host log is written in launch order, so the i-th logged launch is
the i-th matching dispatch of the trace after the warm-up marker (warm-up launches are logged too and dropped by count).
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPI = {0: "none", 1: "bias", 2: "bias+gelu", 3: "gelu'", 4: "add", 5: "scale_rc", 6: "bias+gelu+g'", 7: "mul"}
PAT = re.compile(r"gemm16_|gemm_f32|wgrad_grouped|posconv_direct")


def run(seq_path, steps=6, warmup=3, dtype="f16"):
    import torch
    from bench import synth_batch
    from w2v2_speaker_amd import ops
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # INSTEP_MODEL=large INSTEP_BATCH=32 INSTEP_SAMPLES=80000: BASELINE configs[3]'s per-GPU share instead of configs[1]
    model = os.environ.get("INSTEP_MODEL", "base")
    nb, ns = int(os.environ.get("INSTEP_BATCH", "66")), int(os.environ.get("INSTEP_SAMPLES", "48000"))
    cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-" + model)
    adt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    store = ParamStore(cfg, dev, adt, head="aam", num_speakers=5994, freeze_cnn=True)
    store.init_weights(seed=20211)
    plan = Plan(store, nb, ns, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7)
    trainer = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=max(steps + warmup + 1, 10)),
                             layerdrop_seed=1234, mask_seed=7)
    wav, label = synth_batch(nb, ns, 5994, seed=42133724, device=dev)
    ops.Gemm._log = []
    marks = []
    for i in range(warmup + steps):
        marks.append(len(ops.Gemm._log))
        trainer.train_step(wav, label)
    torch.cuda.synchronize()
    log, ops.Gemm._log = ops.Gemm._log, None
    json.dump({"steps": steps, "warmup": warmup, "marks": marks, "launches": log}, open(seq_path, "w"))
    print(f"logged {len(log)} launches over {warmup}+{steps} steps -> {seq_path}")


def join(seq_path, db_path, out=None):
    seq = json.load(open(seq_path))
    cur = sqlite3.connect(db_path).cursor()
    rows = [(n, s, e) for n, s, e in cur.execute("select name, start, end from kernels order by start") if PAT.search(n)]
    log = seq["launches"]
    if len(rows) != len(log):
        print(f"# WARNING: trace has {len(rows)} matching dispatches, host log {len(log)}: joining the common tail")
    n = min(len(rows), len(log))
    rows, log = rows[len(rows) - n:], log[len(log) - n:]
    first = seq["marks"][seq["warmup"]] - (len(seq["launches"]) - n)
    agg, bad = {}, 0
    for (name, s, e), k in list(zip(rows, log))[max(first, 0):]:
        if k["kernel"].split("_kernel")[0] not in name:
            bad += 1
        if k["kind"] == "gemm":
            key = (k["M"], k["N"], k["K"], EPI[k["epi"]] + ("+aux" if k["aux"] and k["epi"] == 2 else "")
                   + ("+2term" if k["two_term"] else ""), k["kernel"].replace("_kernel", "").replace("gemm16_", "")
                   + (f" x{k['batch']}" if k["batch"] > 1 else ""))
        elif k["kind"] == "posconv":
            key = (k["M"], k["N"], k["K"], "bias+gelu+aux" if k["mode"] == 0 else "add", "posconv_direct x16")
        else:
            key = (k["tokens"], k["problems"], 0, "dW+db", k["kernel"].replace("_kernel", "").replace("wgrad_grouped_", "wgrad_"))
        a = agg.setdefault(key, {"n": 0, "ns": 0, "min": 1 << 62, "max": 0, "flops": 0.0, "alg": 0.0})
        d = e - s
        a["n"] += 1; a["ns"] += d; a["min"] = min(a["min"], d); a["max"] = max(a["max"], d)
        a["flops"] += k["flops"]; a["alg"] += k["alg_flops"]
    steps = seq["steps"]
    tot = sum(a["ns"] for a in agg.values())
    lines = [f"# in-step matrix-core launches by shape: {db_path}; {steps} steps; {tot / 1e6 / steps:.3f} ms/step in these kernels; "
             f"name mismatches in the join: {bad}",
             f"# TF/s = algorithmic FLOPs / duration (two-term K extension NOT credited); TF/s(exec) credits the executed K steps",
             f"{'M':>7} {'N':>5} {'K':>5} {'epilogue':>16} {'kernel':>22} {'n/step':>7} {'avg_us':>8} {'min_us':>8} {'max_us':>8} "
             f"{'ms/step':>8} {'TF/s':>7} {'TF/s(exec)':>10}"]
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        M, N, K, epi, kern = key
        lines.append(f"{M:7d} {N:5d} {K:5d} {epi:>16} {kern:>22} {a['n'] / steps:7.2f} {a['ns'] / a['n'] / 1e3:8.1f} "
                     f"{a['min'] / 1e3:8.1f} {a['max'] / 1e3:8.1f} {a['ns'] / 1e6 / steps:8.3f} "
                     f"{a['alg'] / a['ns'] / 1e3:7.0f} {a['flops'] / a['ns'] / 1e3:10.0f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        st = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
        run(sys.argv[2], steps=st)
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
