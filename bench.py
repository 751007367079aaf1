#!/usr/bin/env python3
"""bench.py -- utterances/sec of one full training step of the hot path (fwd + bwd + gradient
all-reduce + fused Adam) on synthetic 3 s utterances: wav2vec2-base + mean+std pooling + AAM-softmax,
batch 66 per GPU, bf16 activations/weights with f32 accumulation, f32 master weights and optimiser
(BASELINE.json configs[1]; the reference's defaults: frozen CNN, dropout 0.1, LayerDrop 0.05,
SpecAugment time masks).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the forward-layout MFMA GEMM
`gemm_bf16_glds3_kernel<bf16>`): algorithmic FLOPs of its launches divided
by their HIP-event-measured durations inside the timed region.  `cpu_baseline` times the CPU oracle
(oracle/w2v2_oracle.py, kind "port") on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak


def synth_batch(batch, n_samples, num_speakers, seed, device):
    """N(0,1) waveform, per-utterance (x-mean)/(std+1e-5) with the unbiased std
    (ref: src/data/preprocess/input_normalisation.py:54-67); labels randint(0, C)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    wav = torch.randn(batch, n_samples, generator=g)
    std, mean = torch.std_mean(wav, dim=1, keepdim=True)
    wav = (wav - mean) / (std + 1e-5)
    label = torch.randint(0, num_speakers, (batch,), generator=g)
    return wav.to(device), label.to(device)


def cpu_baseline(batch=2, n_samples=48000, num_speakers=5994):
    """The CPU oracle on a bounded sample of the same workload: `batch` utterances of 3 s through
    w2v2-base + mean+std + AAM, forward + backward (CNN frozen) + Adam, f32, all host cores."""
    from oracle import w2v2_oracle as O
    # torch's CPU kernels stop scaling (and then collapse) beyond a few dozen threads at these sizes:
    # use up to 32 host cores and report that number
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cores = torch.get_num_threads()
    cfg = O.OracleConfig.base()
    sd = O.make_state_dict(cfg, 20211)
    W = O.synth_tensor("loss_fn.fc_weights", (num_speakers, 2 * cfg.hidden_size), 20211)
    train = {k: v.requires_grad_(not k.startswith("feature_extractor")) for k, v in sd.items()}
    W.requires_grad_(True)
    wav, label = O.synth_batch(batch, n_samples, num_speakers, seed=42133724)
    params = [v for k, v in train.items() if v.requires_grad] + [W]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]

    def step(i):
        for p in params:
            p.grad = None
        emb = O.speaker_embedding(wav, train, cfg, "mean+std")
        loss, _ = O.aam_softmax(emb, W, label, 0.2, 30.0)
        loss.backward()
        with torch.no_grad():
            for p, (m, v) in zip(params, state):
                if p.grad is not None:
                    O.adam_step(p, p.grad, m, v, i + 1, 1e-5, 0.9)
        return float(loss)

    t0 = time.perf_counter()
    step(0)                                   # warm-up (thread pools, allocator)
    warm = time.perf_counter() - t0
    n = 2 if warm < 12 else 1                 # bound the sample to ~10-30 s of CPU work
    t0 = time.perf_counter()
    for i in range(n):
        step(i + 1)
    dt = (time.perf_counter() - t0) / n
    return {"value": round(batch / dt, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"{n} timed step(s) of {batch} x 3 s utterances, w2v2-base + mean+std + AAM(5994), "
                      f"fwd+bwd (CNN frozen) + Adam, f32 torch CPU oracle ({dt:.2f} s/step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=66, help="utterances per GPU (paper batch size)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--speakers", type=int, default=5994)
    ap.add_argument("--dtype", default="bf16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--pooling", default="mean+std", choices=["mean+std", "attentive", "first+cls"],
                    help="mean+std = the metric's workload; attentive = BASELINE configs[2]")
    ap.add_argument("--model", default="base", choices=["base", "large"],
                    help="base = BASELINE configs[1] (the metric's workload); large = configs[3] geometry "
                         "(24 layers, H=1024; use --seconds 5 --batch 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regularisation", action="store_true", help="dropout/LayerDrop/mask off")
    ap.add_argument("--unfreeze-cnn", action="store_true",
                    help="completely_freeze_feature_extractor=False ablation (127.2 GFLOP/utt)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    from w2v2_speaker_amd import ops
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import SpeakerTrainer

    cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-" + args.model)
    n_samples = int(round(args.seconds * 16000))
    adt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    store = ParamStore(cfg, dev, adt, head="aam", num_speakers=args.speakers, freeze_cnn=not args.unfreeze_cnn,
                       attentive_pool=args.pooling == "attentive",
                       embed_dim=cfg.hidden_size * (1 if args.pooling == "first+cls" else 2))
    store.init_weights(seed=20211)            # identical replicas on every rank (DDP broadcast equivalent)
    reg = Wav2Vec2RegularisationConfig()
    if args.no_regularisation:
        reg = Wav2Vec2RegularisationConfig(attention_dropout=0.0, feat_proj_dropout=0.0, hidden_dropout=0.0,
                                           layerdrop=0.0, mask_time_prob=0.0)
    plan = Plan(store, args.batch, n_samples, train=True, reg=reg, seed=7 + rank, pooling=args.pooling,
                insert_cls_token=args.pooling == "first+cls")
    total = args.steps + args.warmup + 1
    trainer = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=max(total, 10)),
                             layerdrop_seed=1234 + rank, mask_seed=7 + rank)
    wav, label = synth_batch(args.batch, n_samples, args.speakers, seed=42133724 + rank, device=dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(wav, label)
    sync()
    # dominant kernel = the 256x128 3-stage LDS-DMA GEMM (encoder forward products + all data gradients)
    ops.Gemm.profile_begin(lambda g: g.kernel_name in ("gemm_bf16_glds3_kernel", "gemm_bf16_glds4_kernel"))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.train_step(wav, label)
    sync()
    elapsed = time.perf_counter() - t0
    prof = ops.Gemm.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        utt = args.batch * world * args.steps
        fl = cfg.flops_per_utt(n_samples, args.speakers)
        out = {
            "metric": (f"utterances/sec (w2v2-{args.model} + {args.pooling} + AAM-softmax training step, "
                       f"{args.seconds:g} s clips)"),
            "value": round(utt / elapsed, 2), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"wav2vec2-{args.model} + AAM-softmax({args.speakers}), {args.pooling} pooling, "
                                   f"{args.seconds:g} s synthetic audio, "
                                   f"bs={args.batch} per GPU, CNN {'trainable' if args.unfreeze_cnn else 'frozen'}, "
                                   "fwd+bwd+all-reduce+Adam (BASELINE "
                                   + ("configs[1])" if args.model == "base" else "configs[3] geometry)"),
                       "global_batch": args.batch * world, "samples_per_utt": n_samples,
                       "parallelism": f"dp{world}", "regularisation": not args.no_regularisation,
                       "final_loss": round(float(loss), 4)},
            "loss_scale": (None if store.scaler is None else
                           {"scale": float(store.scaler[0]), "skipped_steps": int(store.scaler[3])}),
            "utt_per_sec_per_gpu": round(utt / elapsed / world, 2),
            "model_tflops_per_gpu": round(fl["train_full" if args.unfreeze_cnn else "train_frozen_cnn"] * utt / elapsed
                                          / world / 1e12, 2),
        }
        if prof["launches"]:
            # the two persistent LDS-DMA ring GEMM kernels; the roofline entry is the one with the larger total time
            desc = {"gemm_bf16_glds3_kernel": "256x128x64 3-stage LDS-DMA ring MFMA GEMM: conv4-6, projection, QKV, "
                                              "out-proj, FFN2 forward + the N<=2304 data-gradient products",
                    "gemm_bf16_glds4_kernel": "256x256x32 4-stage LDS-DMA ring MFMA GEMM: conv1-3, FFN1 forward, dH"}
            pmc = {}
            try:     # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py)
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))["kernels"]
            except Exception:
                pass

            def entry(name, k):
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                t = pmc.get(name + "<unsigned short>", {}).get("hbm_bytes_per_launch")
                return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": t,
                        "kernel": f"{name}<bf16> ({desc[name]})", "launches": k["launches"],
                        "ms_per_step": round(k["ms"] / args.steps, 3),
                        "avg_us": round(1e3 * k["ms"] / k["launches"], 2),
                        "avg_gflop_per_launch": round(k["flops"] / k["launches"] / 1e9, 3),
                        "avg_algorithmic_mb_per_launch": round(k["bytes"] / k["launches"] / 1e6, 1)}
            ranked = sorted(prof["by_kernel"].items(), key=lambda kv: -kv[1]["ms"])
            out["roofline"] = entry(*ranked[0])
            if len(ranked) > 1:
                out["roofline_second_kernel"] = entry(*ranked[1])
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
