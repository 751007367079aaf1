"""ECAPA-TDNN training step (BASELINE configs[4]: fbank-like [B, 300, 40] input -> 192-d embedding -> AAM(5994)):
utterances/sec plus an HBM roofline of the elementwise / reduction kernel families of csrc/tdnn.hip.

    python bench.py --model ecapa [--gpus N --steps K --warmup W --dtype bf16|f32]      (bench.py calls bench_ecapa)
    python tools/ecapa_bench.py [--batch 66 --frames 300 --steps 10 --warmup 3]         (1 GPU, same line)

Most of this step is HBM-bound work over channels-last [B*T, C] activations (BatchNorm statistics / apply and their
backward, the SE gate, reflect im2col / col2im, Res2Net adds): the `roofline` entry is the BatchNorm family (largest
HBM-bound share), `roofline_families` lists the others; the MFMA GEMMs are reported beside them.  A family's time is
measured live with HIP events around its launches on the launch stream; `achieved` = algorithmic bytes / that time:

    bn_fwd   3 M C e      two passes are inherent (the statistics need every row before any row is normalised):
                          read a, read a, write y                                      (e = bytes per element)
    bn_bwd   5 M C e      read dy + a for the two column sums, read dy + a again, write da
    se_scale 2 M C e, se_bwd_gate 2 M C e, se_bwd_x 2 M C e,  im2col (1 + k) M Cin e,  col2im (k + 1) M Cin e,
    add_strided 3 M C e

`traffic` = measured HBM bytes per launch from the PMC passes committed in profiles/ (null without them)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_PEAK_TFLOPS = 2500.0
import glob as _glob
PMC_FILE = (sorted(_glob.glob(os.path.join(ROOT, "profiles", "r0?_ecapa_pmc_counters.json"))) or
            [os.path.join(ROOT, "profiles", "r04_ecapa_pmc_counters.json")])[-1]       # newest committed PMC pass


FAMILY_KERNELS = {
    "batchnorm": ("bn_partial_kernel", "bn_apply_kernel", "bn_bwd_partial_kernel", "bn_bwd_apply_kernel"),
    "se_gate": ("se_scale_kernel", "se_bwd_gate_kernel", "se_bwd_x_kernel"),
    "im2col": ("im2col_reflect_kernel", "col2im_reflect_kernel"),
    "res2net_add": ("add_strided_kernel",),
}
GROUP_CLOSERS = dict(FAMILY_KERNELS, batchnorm=("bn_apply_kernel", "bn_bwd_apply_kernel"))


class _FamilyTimer:
    """HIP-event timing of selected ops.* entry points (looked up as attributes of the ops module by the plan)."""

    def __init__(self, ops):
        self.ops, self.saved, self.events, self.on = ops, {}, [], False

    def wrap(self, name, family, nbytes):
        fn = getattr(self.ops, name)
        self.saved[name] = fn

        def timed(*a, **kw):
            if not self.on:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **kw)
            e1.record()
            self.events.append((family, e0, e1, float(nbytes(*a, **kw))))
        setattr(self.ops, name, timed)

    def restore(self):
        for n, fn in self.saved.items():
            setattr(self.ops, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, e0, e1, nb in self.events:
            k = out.setdefault(fam, {"launch_groups": 0, "ms": 0.0, "bytes": 0.0})
            k["launch_groups"] += 1
            k["ms"] += e0.elapsed_time(e1)
            k["bytes"] += nb
        return out


def bench_ecapa(args, world, rank, dev, dist, emit=True):
    """emit=False: return the line's dict instead of printing it (bench.py `also.ecapa_f32_b66`)."""
    from w2v2_speaker_amd import ops
    from w2v2_speaker_amd.ecapa import EcapaConfig, EcapaPlan, EcapaStore, EcapaTrainer
    from w2v2_speaker_amd.optim.schedule import OneCycle

    frames = getattr(args, "frames", 300)
    # the reference trains this model at `precision: 32` (config/experiment/speaker_ecapa_tdnn.yaml:18; BASELINE
    # configs[4] "MFMA off" = no reduced-precision matrix path): f32 is the configs[4] line (exact-f32 products on
    # v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain); bf16 is the faster option, measured separately
    dtype = args.dtype if args.dtype in ("bf16", "f32") else "f32"
    cfg = EcapaConfig()
    st = EcapaStore(cfg, dev, torch.bfloat16 if dtype == "bf16" else torch.float32, num_speakers=5994)
    st.init_weights(1)
    plan = EcapaPlan(st, args.batch, frames, train=True)
    pg = dist.group.WORLD if world > 1 else None
    tr = EcapaTrainer(st, plan, OneCycle(max_lr=1e-3, total_steps=2 * args.steps + args.warmup + 10), pg)
    g = torch.Generator().manual_seed(rank)
    # NB distinct synthetic minibatches, visited round-robin (one fixed batch of 66 random utterances is memorised by
    # the 20.8 M-parameter model within ~25 Adam steps: the round-2 line reported final_loss 0.0 for that reason)
    NB = 8
    feats = torch.randn(NB, args.batch, frames, cfg.input_mel_coefficients, generator=g).to(dev)
    labels = torch.randint(0, 5994, (NB, args.batch), generator=g).to(dev)
    it = [0]

    def next_batch():
        i = it[0] % NB
        it[0] += 1
        return feats[i], labels[i]

    esz = lambda t: t.element_size()
    ft = _FamilyTimer(ops)
    ft.wrap("bn_fwd", "batchnorm", lambda a, lda, work, mr, run, ga, be, y, ldy, M, C, *r: 3 * M * C * esz(a))
    ft.wrap("bn_bwd", "batchnorm", lambda dy, lddy, a, lda, mr, ga, work, dga, dbe, da, ldda, M, C, *r, **kw:
            (7 if kw.get("dy2") is not None else 5) * M * C * esz(a))
    ft.wrap("se_scale", "se_gate", lambda x, g_, y, B, T, C: 2 * B * T * C * esz(x))
    ft.wrap("se_bwd_gate", "se_gate", lambda d, x, dg, B, T, C: 2 * B * T * C * esz(x))
    ft.wrap("se_bwd_x", "se_gate", lambda d, g_, ds, dx, B, T, C: 2 * B * T * C * esz(d))
    ft.wrap("im2col_reflect", "im2col",
            lambda x, ldx, col, B, T, Cin, k, dil, x2=None, ldx2=0: ((2 if x2 is not None else 1) + k) * B * T * Cin * esz(x))
    ft.wrap("col2im_reflect", "im2col", lambda dcol, dx, lddx, B, T, Cin, k, dil, acc: (k + 1 + int(acc)) * B * T * Cin * esz(dx))
    ft.wrap("add_strided", "res2net_add", lambda a, lda, b, ldb, y, ldy, M, C: (3 if b is not None else 2) * M * C * esz(a))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.train_step(*next_batch())
    sync()
    # pass 1: the timed region of the metric, no per-launch events (they serialise the host against the stream)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.train_step(*next_batch())
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # pass 2 (not part of `value`): the same steps with HIP events around the HBM-bound families and the GEMMs
    psteps = 0 if os.environ.get("W2V2_BENCH_NO_FAMILY_PASS") else min(args.steps, 5)
    ft.on = True
    ops.Gemm.profile_begin(lambda gm: True)
    for _ in range(psteps):
        tr.train_step(*next_batch())
    gp = ops.Gemm.profile_end()
    fam = ft.summary()
    ft.restore()
    if rank != 0:
        return
    T, C = frames, cfg.channels
    w = C[1] // cfg.res2net_scale
    fwd = 2 * T * (cfg.input_mel_coefficients * cfg.kernel_sizes[0] * C[0]
                   + 3 * (2 * C[1] * C[1] + (cfg.res2net_scale - 1) * w * w * 3) + C[-1] * C[-1]
                   + C[-1] * cfg.attention_channels * 2) + 2 * (2 * C[-1] * cfg.lin_neurons + cfg.lin_neurons * 5994)
    pmc, pmc_stale = {}, None
    try:     # stale = the kernel sources have changed since the PMC passes were taken (hash recorded by pmc_counters.py)
        from w2v2_speaker_amd._build import source_hash
        pmc = json.load(open(PMC_FILE))
        pmc_stale = pmc.get("source_hash") != source_hash()
    except Exception:
        pass

    def traffic(name):
        """Measured HBM bytes per launch group: the PMC bytes of the family's kernels over the profiled run, divided
        by the launches of the kernel that closes a group (bn_fwd = partial + apply: one group)."""
        kern = pmc.get("kernels", {})
        tot = groups = 0
        for kname, rec in kern.items():
            base = kname.split("<")[0]
            if base in FAMILY_KERNELS[name]:
                tot += rec["hbm_bytes_per_launch"] * rec["launches"]
                if base in GROUP_CLOSERS[name]:
                    groups += rec["launches"]
        return int(tot / groups) if groups else None

    def hbm_entry(name, k):
        ach = k["bytes"] / (k["ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic(name),
                "pmc_source": os.path.relpath(PMC_FILE, ROOT), "pmc_stale": pmc_stale,
                "family": name, "launch_groups_per_step": k["launch_groups"] // psteps,
                "ms_per_step": round(k["ms"] / psteps, 3),
                "avg_us": round(1e3 * k["ms"] / k["launch_groups"], 2),
                "avg_algorithmic_mb_per_launch_group": round(k["bytes"] / k["launch_groups"] / 1e6, 2)}
    ranked = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])
    utt = args.batch * world * args.steps
    out = {"metric": "utterances/sec (ECAPA-TDNN C=1024 + AAM-softmax training step, 300 filterbank frames)",
           "value": round(utt / elapsed, 1), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": dtype,
           "mfma": ("f32 exact: products on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-for-bit an fmaf chain, f32 "
                    "vector RATE) -- BASELINE configs[4] 'MFMA off' read as 'no reduced-precision matrix path'"
                    if dtype == "f32" else "bf16 operands on v_mfma_f32_16x16x32_bf16, f32 accumulation"),
           "data": f"synthetic ({NB} distinct minibatches, round-robin)",
           "config": {"workload": f"ECAPA-TDNN (C=1024, 5 blocks, attentive statistics pooling, 192-d) + AAM-softmax(5994), "
                                  f"[{args.batch}, {frames}, 40] synthetic filterbank frames per GPU, fwd+bwd+"
                                  "all-reduce+Adam (BASELINE configs[4])",
                      "global_batch": args.batch * world, "parallelism": f"dp{world}",
                      "final_loss": round(float(loss), 4)},
           "fwd_gflop_per_utt": round(fwd / 1e9, 3),
           "model_tflops_per_gpu": round(3 * fwd * utt / elapsed / world / 1e12, 1)}
    if ranked and psteps:
        out["roofline"] = hbm_entry(*ranked[0])
        out["roofline_families"] = [hbm_entry(*kv) for kv in ranked[1:]]
    if gp["launches"] and psteps:
        ach = gp["flops"] / (gp["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS if dtype == "bf16" else 157.3        # f32-input MFMA = the f32 vector rate
        out["gemm_mfma"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                            "frac": round(ach / peak, 4), "launches_per_step": gp["launches"] // psteps,
                            "ms_per_step": round(gp["ms"] / psteps, 3),
                            "by_kernel": {k: {"ms_per_step": round(v["ms"] / psteps, 3), "launches_per_step": v["launches"] // psteps,
                                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                          for k, v in sorted(gp["by_kernel"].items(), key=lambda kv: -kv[1]["ms"])}}
    if not emit:
        return out
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=66)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="f32", choices=["bf16", "f32"])
    a = ap.parse_args()
    torch.cuda.set_device(0)
    bench_ecapa(a, 1, 0, torch.device("cuda", 0), None)
