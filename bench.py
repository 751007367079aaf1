#!/usr/bin/env python3
"""bench.py -- utterances/sec of one full training step of the hot path (fwd + bwd + gradient
all-reduce + fused Adam) on synthetic 3 s utterances: wav2vec2-base + mean+std pooling + AAM-softmax,
batch 66 per GPU (BASELINE.json configs[1]; the reference's defaults: frozen CNN, dropout 0.1, LayerDrop 0.05,
SpecAugment time masks).  Arithmetic: fp16 activations / weight copies on the matrix cores with f32
accumulation, f32 master weights and optimiser, dynamic loss scaling -- the reference's own precision
(PL `precision: 16`, config/experiment/speaker_wav2vec2_aam.yaml:17), the mode whose embeddings sit within
1e-3 rel-L2 of the f32 reference (tests/test_parity_gpu.py).  `--dtype bf16|f32` select the other modes.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or plainly as above, in which case this process starts N fresh children -- one per GPU,
before it has touched a GPU itself -- and waits for them.  One rank per GPU over RCCL (torch.distributed "nccl").

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel of the step (largest total time among the
MFMA GEMM kernels): algorithmic FLOPs of its launches divided by their durations inside the timed region, measured with
HIP events that carry the dispatch's own begin / end timestamps (w2v2_gemm_timed -> hipExtLaunchKernelGGL: the duration
rocprofv3 reports for the kernel); `traffic` / `mfma_busy` come from the committed rocprofv3 PMC passes of the same command under profiles/.
`cpu_baseline` times the CPU oracle (oracle/w2v2_oracle.py, kind "port") on the host cores with the protocol of
BASELINE.md section 3 (config 1: bs 8, CE head).
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# kernel arguments in device memory (w2v2_speaker_amd/__init__.py explains; -3 % step time): set here as well, before
# torch can initialise the HIP runtime in this process or in the ranks it spawns
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch

MFMA_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
def _latest(pattern, default):
    """Newest committed evidence file of a kind (profiles/rNN_...): rounds add files, they never rewrite old ones."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return c[-1] if c else os.path.join(ROOT, "profiles", default)


PMC_FILE = _latest("r0?_pmc_counters.json", "r04_pmc_counters.json")   # tools/profile_round.sh -> tools/pmc_counters.py


def synth_batch(batch, n_samples, num_speakers, seed, device):
    """N(0,1) waveform, per-utterance (x-mean)/(std+1e-5) with the unbiased std
    (ref: src/data/preprocess/input_normalisation.py:54-67); labels randint(0, C)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    wav = torch.randn(batch, n_samples, generator=g)
    std, mean = torch.std_mean(wav, dim=1, keepdim=True)
    wav = (wav - mean) / (std + 1e-5)
    label = torch.randint(0, num_speakers, (batch,), generator=g)
    return wav.to(device), label.to(device)


def cpu_baseline(batch=8, n_samples=48000, num_speakers=1211):
    """the CPU oracle on BASELINE config 1 -- w2v2-base + CE head (C = 1211), batch 8 x 3 s,
    f32, forward + backward (CNN frozen) + Adam; 2 warm-up + 5 timed steps, median utt/s, at the thread count that a
    short sweep over {8, 16, 32, 64} finds fastest (torch's CPU kernels stop scaling beyond a few dozen threads and
    lose to oversubscription on a busy host); `cores` = the threads actually used, the sweep is reported."""
    from oracle import w2v2_oracle as O
    cfg = O.OracleConfig.base()
    sd = O.make_state_dict(cfg, 20211)
    E = 2 * cfg.hidden_size
    W = O.synth_tensor("fc_list.0.0.weight", (num_speakers, E), 20211).requires_grad_(True)
    b = O.synth_tensor("fc_list.0.0.bias", (num_speakers,), 20211).requires_grad_(True)
    train = {k: v.requires_grad_(not k.startswith("feature_extractor")) for k, v in sd.items()}
    wav, label = O.synth_batch(batch, n_samples, num_speakers, seed=42133724)
    params = [v for k, v in train.items() if v.requires_grad] + [W, b]
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]

    def step(i):
        for p in params:
            p.grad = None
        emb = O.speaker_embedding(wav, train, cfg, "mean+std")
        loss, _ = O.ce_head(emb, W, b, label)
        loss.backward()
        with torch.no_grad():
            for p, (m, v) in zip(params, state):
                if p.grad is not None:
                    O.adam_step(p, p.grad, m, v, i + 1, 1e-5, 0.9)
        return float(loss.detach())

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        for i in range(warm):
            step(i)
        ts = []
        for i in range(timed):
            t0 = time.perf_counter()
            step(warm + i)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    # the stated baseline is the BEST honest CPU figure, not an oversubscribed one (64 threads gave
    # 2.66 utt/s where 8 gave 7.32): short sweep over thread counts (1 warm-up + 2 timed steps each, bounded), then the
    # 2 warm-up + 5 timed steps of BASELINE.md section 3 at the best count
    ncpu = os.cpu_count() or 1
    sweep, budget_t0 = {}, time.perf_counter()
    for th in (8, 16, 32, 64):
        if th > ncpu and th != 8:
            continue
        if sweep and time.perf_counter() - budget_t0 > 40.0:      # bounded sample: stop sweeping on a slow host
            break
        sweep[th] = run(min(th, ncpu), 1, 2)
    cores = min(sweep, key=sweep.get)
    dt = run(min(cores, ncpu), 2, 5)
    cores = min(cores, ncpu)
    return {"value": round(batch / dt, 4), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "gflops": round(97.73 * batch / dt, 1),
            "thread_sweep_utt_per_sec": {str(k): round(batch / v, 3) for k, v in sweep.items()},
            "sample": f"BASELINE config 1: median of 5 timed steps (2 warm-up) of {batch} x 3 s utterances, w2v2-base + "
                      f"mean+std + CE({num_speakers}), fwd+bwd (CNN frozen) + Adam, f32 torch CPU oracle, {cores} of "
                      f"{ncpu} host threads = the fastest of the sweep {sorted(sweep)} ({dt:.2f} s/step)"}


def tail_families(trainer, plan, store, wav, label, steps=4):
    """the NON-GEMM tail of the step against its own rooflines, on the driver line.  After the timed
    region, `steps` more training steps run with HIP events around every launch of the tail's families (LayerNorm forward /
    backward, the three attention kernels, Adam, the layer-0 convolution): per family the launches per step, the mean
    duration, the ALGORITHMIC bytes per launch (operands read once, results written once) and achieved / 8 TB/s.  The
    attention kernels are VALU-bound, not HBM-bound: their entry also carries the matrix TFLOP/s (fwd 4 T^2 d, bwd 10 T^2 d per
    head) and the share of computed scores that are real (padding of T to the 16-row / 16-key blocks the geometry walks).
    The events sit between launches of one stream: a bracket also holds the event records themselves, so the median of 32
    EMPTY brackets is measured in the same pass and subtracted (both figures are on the line).  Against the rocprofv3 kernel
    trace of the same command (profiles/r06_bench_b66_kernel_stats.txt) the corrected durations agree within ~2 us (LayerNorm
    forward 10.3 vs 12.1, attention forward 23.0 vs 23.3, backward 57.9 vs 58.3): good enough to rank the tail against its
    roofline on the driver's own box, not a replacement for the trace."""
    import w2v2_speaker_amd.ops as O_
    rec, orig = [], {}

    def wrap(name, family, nbytes, extra=None):
        fn = getattr(O_, name)
        orig[name] = fn

        def timed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            rec.append((family, e0, e1, float(nbytes(*a, **kw)), extra(*a, **kw) if extra else None))
            return out
        setattr(O_, name, timed)

    esz = lambda t: t.element_size()
    wrap("layernorm_fwd", "layernorm_fwd", lambda x, r, g, b, y, *a, **k: x.numel() * esz(x) * (4 if r is not None else 2))
    wrap("layernorm_bwd", "layernorm_bwd", lambda dy, s_, m, rs, g, ds, d_r, *a, **k: dy.numel() * esz(dy) * (4 if d_r is not None else 3))
    wrap("attention_fwd", "attention_fwd", lambda qkv, ctx, lse, *a, **k: qkv.numel() * esz(qkv) + ctx.numel() * esz(ctx) + lse.numel() * 4,
         lambda qkv, ctx, lse, B, T, heads, d, *a, **k: (4.0 * B * heads * T * T * d, T))
    wrap("attention_bwd", "attention_bwd", lambda qkv, ctx, dctx, lse, dqkv, delta, *a, **k: (2 * qkv.numel() + 2 * ctx.numel()) * esz(qkv) * 1.0 + 3 * lse.numel() * 4,
         lambda qkv, ctx, dctx, lse, dqkv, delta, B, T, heads, d, *a, **k: (10.0 * B * heads * T * T * d, T))
    wrap("adam_step", "adam", lambda p_, g_, m, v, pb, n, *a, **k: n * (28.0 + (pb.element_size() if pb is not None else 0)))
    wrap("conv0_groupnorm_gelu", "conv0", lambda wav_, w, ga, be, y, *a, **k: 2 * wav_.numel() * 4 + y.numel() * esz(y))
    try:
        for _ in range(steps):
            trainer.train_step(wav, label)
        torch.cuda.synchronize()
    finally:
        for n, fn in orig.items():
            setattr(O_, n, fn)
    # what a bracket costs by itself: event pairs around NOTHING, between two kernels of the same stream
    empt = []
    z = torch.zeros(1 << 20, device=wav.device)
    for _ in range(32):
        z.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        empt.append((e0, e1))
        z.add_(1.0)
    torch.cuda.synchronize()
    overhead = float(np.median([a.elapsed_time(b) * 1e3 for a, b in empt]))
    fam = {}
    for family, e0, e1, nb, extra in rec:
        f = fam.setdefault(family, {"us": 0.0, "bytes": 0.0, "n": 0, "flops": 0.0, "T": None})
        f["us"] += e0.elapsed_time(e1) * 1e3
        f["bytes"] += nb
        f["n"] += 1
        if extra:
            f["flops"] += extra[0]
            f["T"] = extra[1]
    out = []
    for family, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
        raw_avg = f["us"] / f["n"]
        avg = max(raw_avg - overhead, 0.5)               # the kernel's share of the bracket
        f["us"] = avg * f["n"]
        gbs = f["bytes"] / f["n"] / (avg * 1e-6) / 1e9
        e = {"family": family, "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
             "frac": round(gbs / 8000.0, 4), "launches_per_step": round(f["n"] / steps, 2), "avg_us": round(avg, 2),
             "ms_per_step": round(f["us"] / steps / 1e3, 4), "algorithmic_mb_per_launch": round(f["bytes"] / f["n"] / 1e6, 2),
             "bracket_us": round(raw_avg, 2), "empty_bracket_us": round(overhead, 2)}
        if f["flops"]:
            T = f["T"]
            pad = -(-T // 32) * 32 if (-(-T // 32) * 32 < -(-T // 64) * 64 and T <= 512) else -(-T // 64) * 64
            pad16 = -(-T // 16) * 16
            e.update({"bound_in_practice": "valu (softmax + dropout arithmetic; profiles/r05_attention_pmc.txt)",
                      "matrix_tflops": round(f["flops"] / f["us"] / 1e6, 1), "matrix_frac_of_2500": round(f["flops"] / f["us"] / 1e6 / 2500.0, 4),
                      "valid_score_fraction": round(T * T / float(pad16 * pad16), 4),
                      "rows_padded_to": pad, "note": "fully padded 16-row / 16-key blocks are skipped: computed scores = "
                                                     f"{pad16} x {pad16} per head for T = {T}"})
        out.append(e)
    return out


def eer_leg(store, dev, dtype_name):
    """The metric's second half, "eval EER" (SURVEY 8d): the fixed synthetic trial list of
    w2v2_speaker_amd/data/synthetic.py (8 speakers x 4 utterances of 3 s, all 496 pairs) embedded by THIS engine in the
    benchmarked precision (eval mode, mean+std), scored like the reference's evaluator (cosine -> (s+1)/2 clip ->
    calculate_eer; ref: src/evaluation/speaker/speaker_recognition_evaluator.py:46-115), next to the REFERENCE's own EER
    on the same waveforms and weights (tests/golden/g12_eer.npz, produced by running the reference: make_goldens.py `eer`).
    Runs after the timed region; the store's weights are replaced by the goldens' name-keyed synthetic ones."""
    from w2v2_speaker_amd.data.synthetic import score_trials, synth_state_dict, synth_trial_set
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.eval_metrics import calculate_eer, calculate_mdc
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "g12_eer.npz"))
        wav, _spk, _keys, trials = synth_trial_set()
        store.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(store.shapes, 20211).items()})
        ev = Plan(store, wav.shape[0], wav.shape[1], train=False)
        e = ev.embed(torch.from_numpy(wav).to(dev)).float().cpu().numpy()
        torch.cuda.synchronize()
        ref = g["embedding"].astype(np.float64)
        per_utt = np.linalg.norm(e - ref, axis=1) / np.linalg.norm(ref, axis=1)
        gt, sc = score_trials(e, trials)
        eer, _ = calculate_eer(gt, sc)
        mdc, _ = calculate_mdc(gt, sc)
        del ev
        # the sharper discriminator: the evaluator's centred + length-normed branch removes the
        # direction all embeddings share (ref: cosine_distance.py:117-127, speaker_recognition_evaluator.py:162-167),
        # fitted -- as a reference run would -- on the embeddings THIS engine produced
        from w2v2_speaker_amd.evaluation.speaker.cosine_distance import CosineDistanceEvaluator, EmbeddingSample, EvaluationPair
        et = torch.from_numpy(e)
        smp = [EmbeddingSample(k, v) for k, v in zip(_keys, et)]
        cev = CosineDistanceEvaluator(True, True, len(smp))
        cev.fit_parameters([v for v in et], [])
        csc = np.clip((np.array(cev._compute_prediction_scores([(smp[i], smp[j]) for _, i, j in trials])) + 1) / 2, 0, 1)
        cres = cev.evaluate([EvaluationPair(bool(s), _keys[i], _keys[j]) for s, i, j in trials], smp)
        centred = {"hip_" + dtype_name: round(float(cres["eer"]), 6), "reference": round(float(g["eer_cl"]), 6),
                   "abs_diff": round(abs(float(cres["eer"]) - float(g["eer_cl"])), 6),
                   "min_dcf_hip": round(float(cres["mdc"]), 5), "min_dcf_reference": round(float(g["mdc_cl"]), 5),
                   "max_abs_score_diff": float(np.abs(csc - g["scores_cl"]).max()),
                   "what": "CosineDistanceEvaluator(center_before_scoring=True, length_norm_before_scoring=True) fitted on the 32 embeddings"}
        return {"hip_" + dtype_name: round(float(eer), 6), "reference": round(float(g["eer"]), 6), "centred": centred,
                "abs_diff": round(abs(float(eer) - float(g["eer"])), 6),
                "min_dcf_hip": round(float(mdc), 5), "min_dcf_reference": round(float(g["mdc"]), 5),
                "trials": len(trials), "target_trials": int(sum(gt)),
                "max_abs_score_diff": float(np.abs(np.array(sc) - g["scores"]).max()),
                "embedding_rel_l2_per_utterance_max": float(per_utt.max()),
                "source": "tests/golden/g12_eer.npz (reference embeddings / scores / EER of the same 32 synthetic utterances)"}
    except Exception as ex:              # a side leg must never cost the headline line
        return {"error": repr(ex)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=66, help="utterances per GPU (paper batch size)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--speakers", type=int, default=5994)
    ap.add_argument("--dtype", default=None, choices=["f16", "bf16", "f32"],
                    help="default = the reference's own precision for the model: f16 for wav2vec2 (fp16-AMP, embeddings "
                         "within 1e-3 of the f32 reference), f32 for --model ecapa (`precision: 32`); bf16: 8-bit "
                         "significand option; f32: exact mode")
    ap.add_argument("--pooling", default="mean+std", choices=["mean+std", "attentive", "first+cls"],
                    help="mean+std = the metric's workload; attentive = BASELINE configs[2]")
    ap.add_argument("--model", default="base", choices=["base", "large", "ecapa"],
                    help="base = BASELINE configs[1] (the metric's workload); large = configs[3] geometry "
                         "(24 layers, H=1024; use --seconds 5 --batch 32); ecapa = configs[4] (ECAPA-TDNN on 300 x 40 "
                         "filterbank frames, HBM-roofline entry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the tail-family pass (4 extra steps with HIP events, after the timed region)")
    ap.add_argument("--no-eer", action="store_true", help="skip the eval-EER leg (32 synthetic utterances, after the timed region)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short legs reported under `also` (bf16 mode + BASELINE configs[2], [3], [4])")
    ap.add_argument("--no-regularisation", action="store_true", help="dropout / LayerDrop / masks off")
    ap.add_argument("--unfreeze-cnn", action="store_true",
                    help="completely_freeze_feature_extractor=False ablation (127.2 GFLOP/utt)")
    a = ap.parse_args()
    if a.dtype is None:
        a.dtype = "f32" if a.model == "ecapa" else "f16"
    return a


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child(rank: int, world: int, port: int, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.argv = argv
    run(parse_args())


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # self-launch: N fresh processes ("spawn": nothing of this process is inherited), started before this one
        # has made any GPU call; it only waits for them (never an exec of a process that has touched the GPU)
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        port = _free_port()
        procs = [ctx.Process(target=_child, args=(r, args.gpus, port, list(sys.argv))) for r in range(args.gpus)]
        for p in procs:
            p.start()
        # wait for ALL ranks, but give up as soon as ONE fails: its peers would otherwise sit in init_process_group or
        # in a collective until the RCCL / store timeout (10-30 min) while holding their GPUs
        from multiprocessing.connection import wait as wait_any
        rc, alive = 0, {p.sentinel: p for p in procs}
        while alive and rc == 0:
            for s in wait_any(list(alive), timeout=5.0):
                p = alive.pop(s)
                p.join()
                rc = rc or (p.exitcode or 0)
        if rc != 0:
            for p in alive.values():
                p.terminate()
            for p in alive.values():
                p.join(10)
                if p.is_alive():
                    p.kill()
        raise SystemExit(rc)
    run(args)


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("W2V2_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    rccl = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm.  W2V2_DIST_BACKEND=gloo + W2V2_SHARE_GPU=1 exist only to rehearse the N > 1 control flow
        # (self-spawn, bucket events, side stream) on a ONE-GPU box, where RCCL refuses two ranks on one device
        backend = os.environ.get("W2V2_DIST_BACKEND", "nccl")
        # An RCCL workgroup's LDS does not fit beside a 144 KiB GEMM workgroup, so every channel takes a CU away from
        # the persistent GEMM grids for the length of a collective.  The single-round products (234 tiles) keep their
        # one round as long as >= 234 of the 256 CUs stay free: cap the channels at 16 (376 MB per step at the
        # bandwidth of 16 channels is still far shorter than the backward it hides under).  Override via the environment.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # per-rank confirmation that the collective runs on the GPUs over RCCL: every rank contributes rank + 1
        chk = torch.full((1024,), float(rank + 1), device=dev)
        dist.all_reduce(chk)
        torch.cuda.synchronize()
        ok = bool((chk == world * (world + 1) / 2).all())
        flags = [None] * world
        dist.all_gather_object(flags, {"rank": rank, "device": torch.cuda.get_device_name(local_rank), "allreduce_ok": ok})
        rccl = {"backend": dist.get_backend(), "ranks": flags}

    if args.model == "ecapa":
        from tools.ecapa_bench import bench_ecapa
        bench_ecapa(args, world, rank, dev, dist)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

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
    total = args.steps + args.warmup + 1 + 8          # + the tail-family pass behind the timed region (tail_families)
    trainer = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=max(total, 10)),
                             layerdrop_seed=1234 + rank, mask_seed=7 + rank)
    wav, label = synth_batch(args.batch, n_samples, args.speakers, seed=42133724 + rank, device=dev)
    if world > 1:
        # what PL's DDP wrapper does when it wraps the module: every replica starts from rank 0's state,
        # whatever the ranks initialised or loaded themselves
        trainer.broadcast_state(0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(wav, label)
    sync()
    solo_ms = None
    if world > 1:
        # A 1-GPU figure from the SAME invocation (same box, same thermal state): every rank steps alone -- no
        # collective, same kernels -- for K steps; the data-parallel pass below is then read against it
        # (`ddp.step_time_ratio_vs_solo` = solo / ddp step time = what the overlapped all-reduce costs; the driver
        # computes the scaling efficiency proper from its own 1-GPU run).  Replicas are re-synchronised afterwards.
        class _Solo:
            world = 1
            def bucket_ready(self, name): pass
            def wait(self): pass
        solo = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=max(total, 10)),
                              layerdrop_seed=4321, mask_seed=99, reducer=_Solo())
        for _ in range(2):
            solo.train_step(wav, label)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            solo.train_step(wav, label)
        torch.cuda.synchronize()
        solo_ms = 1e3 * (time.perf_counter() - ts) / args.steps
        trainer.broadcast_state(0)
        sync()
    skipped0 = int(store.scaler[3]) if store.scaler is not None else 0
    ring = ("gemm16_ring_256x128_kernel", "gemm16_phased_256x256_kernel")
    ops.Gemm.profile_begin(lambda g: g.kernel_name in ring)
    n_skip_layers = 0
    # SURVEY 8(d) timing protocol: besides the wall clock over the K steps (the figure `value` is computed from), one HIP
    # event per step boundary on the compute stream -> per-step device times, reported as median / p10 / p90
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_skips = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, _ = trainer.train_step(wav, label)
        marks[i + 1].record()
        n_skip_layers += len(plan._skip)
        step_skips.append(len(plan._skip))
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    prof = ops.Gemm.profile_end()
    ddp = None
    if world > 1:
        mine = torch.tensor([elapsed, float(np.median(step_ms)), solo_ms, float(n_skip_layers)], dtype=torch.float64,
                            device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        elapsed = float(allr[:, 0].max())                    # the contract: MAX over ranks
        ddp = {"per_rank_ms_per_step": [round(1e3 * v / args.steps, 3) for v in allr[:, 0]],
               "per_rank_step_ms_median": [round(v, 3) for v in allr[:, 1]],
               "per_rank_solo_ms_per_step": [round(v, 3) for v in allr[:, 2]],
               "per_rank_layerdrop_skipped_layers_per_step": [round(v / args.steps, 2) for v in allr[:, 3]],
               "rank_spread_ms": round(1e3 * float(allr[:, 0].max() - allr[:, 0].min()) / args.steps, 3),
               "step_time_ratio_vs_solo": round(float(allr[:, 2].max()) / (1e3 * elapsed / args.steps), 4),
               "note": "solo = the same K steps on every rank without the collective, same invocation; LayerDrop draws "
                       "differ per rank and per pass (reported), so the ratio is indicative to about +-1 %"}
    if rank == 0:
        utt = args.batch * world * args.steps
        fl = cfg.flops_per_utt(n_samples, args.speakers)
        raw = fl["train_full" if args.unfreeze_cnn else "train_frozen_cnn"]
        # LayerDrop: a skipped layer does no forward and no backward work (3 x its forward FLOPs); the expectation is
        # p = 0.05 of the encoder, the figure below uses the layers this rank actually skipped in the timed steps
        done = raw - 3.0 * fl["layer"] * n_skip_layers / args.steps
        out = {
            "metric": (f"utterances/sec (w2v2-{args.model} + {args.pooling} + AAM-softmax training step, "
                       f"{args.seconds:g} s clips)"),
            "value": round(utt / elapsed, 2), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_step_median": round(float(np.median(step_ms)), 3),
            "ms_per_step_p10": round(float(np.percentile(step_ms, 10)), 3),
            "ms_per_step_p90": round(float(np.percentile(step_ms, 90)), 3),
            "ms_per_step_no_layerdrop_median": (round(float(np.median(step_ms[np.array(step_skips) == 0])), 3)
                                                if any(k == 0 for k in step_skips) else None),
            "step_timing": "HIP events on the compute stream at every step boundary (rank 0); ms_per_step = wall clock / K",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"wav2vec2-{args.model} + AAM-softmax({args.speakers}), {args.pooling} pooling, "
                                   f"{args.seconds:g} s synthetic audio, "
                                   f"bs={args.batch} per GPU, CNN {'trainable' if args.unfreeze_cnn else 'frozen'}, "
                                   "fwd+bwd+all-reduce+Adam (BASELINE "
                                   + ("configs[1])" if args.model == "base" else "configs[3] geometry)"),
                       "global_batch": args.batch * world, "samples_per_utt": n_samples,
                       "parallelism": f"dp{world}", "regularisation": not args.no_regularisation,
                       "final_loss": round(float(loss), 4),
                       "precision": {"f16": "fp16 operands (two-term fp16 weights on the value / output projections), f32 "
                                            "accumulation, f32 master weights + Adam, dynamic loss scale",
                                     "bf16": "bf16 operands, f32 accumulation, f32 master weights + Adam",
                                     "f32": "exact f32"}[args.dtype]},
            "loss_scale": (None if store.scaler is None else
                           {"scale": float(store.scaler[0]),
                            "skipped_steps_in_timed_region": int(store.scaler[3]) - skipped0}),
            "utt_per_sec_per_gpu": round(utt / elapsed / world, 2),
            "model_tflops_per_gpu": round(raw * utt / elapsed / world / 1e12, 2),
            "model_tflops_per_gpu_layerdrop_adjusted": round(done * utt / elapsed / world / 1e12, 2),
            "layerdrop_skipped_layers_per_step": round(n_skip_layers / args.steps, 3),
        }
        if rccl is not None:
            out["rccl"] = rccl
        if ddp is not None:
            out["ddp"] = ddp
        if prof["launches"]:
            desc = {"gemm16_ring_256x128_kernel": "256x128x64 3-stage LDS-DMA ring MFMA GEMM: conv4-6, projection, QKV, "
                                              "out-proj, FFN2 forward + the N<=2304 data-gradient products",
                    "gemm16_phased_256x256_kernel": "256x256x64 phased LDS-DMA MFMA GEMM (two wave groups in anti-phase): conv1-3, "
                                      "FFN1 forward, dH"}
            pmc, pmc_stale = {}, None
            try:     # HBM bytes per launch and matrix-pipe busy fraction from the committed PMC passes of this command
                # (separate rocprofv3 --pmc runs: a timed run cannot carry counters); stale = the kernel sources have
                # changed since those passes were taken (source hash recorded by tools/pmc_counters.py)
                from w2v2_speaker_amd._build import source_hash
                rec = json.load(open(PMC_FILE))
                pmc = rec["kernels"]
                pmc_stale = rec.get("source_hash") != source_hash()
            except Exception:
                pass
            tsym = {"f16": "_Float16", "bf16": "unsigned short"}.get(args.dtype, "")
            # what the chip sustains on NOTHING BUT register-resident 16-bit MFMAs with random operands on all CUs
            # (tools/probes/mfma_sustained_probe, committed output): the power-limited ceiling under `peak`
            sustained = None
            try:
                for ln in open(_latest("r0?_mfma_sustained.txt", "r04_mfma_sustained.txt")):
                    f = ln.split()
                    if len(f) == 5 and f[0] == "256" and f[1] == "1" and f[2] == "random":
                        sustained = float(f[4])       # first block of the file = v_mfma_f32_16x16x32_f16 (the kernels' shape)
                        break
            except Exception:
                pass

            def entry(name, k):
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                rec = pmc.get(f"{name}<{tsym}, {tsym}>") or pmc.get(f"{name}<{tsym}, {tsym}, 0>") or {}
                # matrix-pipe utilisation in EXECUTED flops: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) x 1024
                # FLOP per busy cycle (a v_mfma_f32_16x16x32 = 16384 FLOP holds its SIMD's pipe for 16 cycles), per launch,
                # over THIS run's measured launch duration and the nominal peak.  It counts the two-term K extension and
                # the padded tile rows the algorithmic figure leaves out, so it cannot fall below `frac` (the round-4
                # field divided by a GRBM_GUI_ACTIVE window that includes per-dispatch overhead and did)
                busy_cyc = rec.get("mfma_busy_cycles_per_launch")
                avg_s = 1e-3 * k["ms"] / k["launches"]
                exec_tf = busy_cyc * 1024.0 / avg_s / 1e12 if busy_cyc else None
                return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": rec.get("hbm_bytes_per_launch"),
                        "mfma_busy": round(exec_tf / MFMA_PEAK_TFLOPS, 4) if exec_tf else None,
                        "mfma_executed_tflops": round(exec_tf, 1) if exec_tf else None,
                        "peak_sustained_measured": sustained,
                        "frac_of_sustained": round(ach / sustained, 4) if sustained else None,
                        "pmc_source": os.path.relpath(PMC_FILE, ROOT), "pmc_stale": pmc_stale,
                        "kernel": f"{name}<{args.dtype}> ({desc[name]})", "launches": k["launches"],
                        "ms_per_step": round(k["ms"] / args.steps, 3),
                        "avg_us": round(1e3 * k["ms"] / k["launches"], 2),
                        "avg_gflop_per_launch": round(k["flops"] / k["launches"] / 1e9, 3),
                        "avg_algorithmic_mb_per_launch": round(k["bytes"] / k["launches"] / 1e6, 1)}
            ranked = sorted(prof["by_kernel"].items(), key=lambda kv: -kv[1]["ms"])
            out["roofline"] = entry(*ranked[0])
            if len(ranked) > 1:
                out["roofline_second_kernel"] = entry(*ranked[1])
        if world == 1 and not args.no_families and not os.environ.get("W2V2_BENCH_NO_FAMILY_PASS"):
            try:
                out["roofline_families"] = tail_families(trainer, plan, store, wav, label)
            except Exception as ex:          # a side leg must never cost the headline line
                out["roofline_families"] = {"error": repr(ex)}
        if world == 1 and args.model == "base" and args.pooling == "mean+std" and not args.no_eer:
            out["eer"] = eer_leg(store, dev, args.dtype)
        if world == 1 and args.dtype == "f16" and args.model == "base" and not args.no_also:
            # Short legs of the same engine on the other BASELINE configurations, so that the driver's line carries them
            # (the headline's step and warm-up counts each; `value` above is untouched by them):
            #   bf16          configs[1] says "bf16"; the headline is fp16 because only fp16 operands keep the embedding
            #                 within the 1e-3 rel-L2 target (tests/test_parity_gpu.py; profiles/r04_parity.json)
            #   attentive_b66 configs[2]: w2v2-base + attentive statistics pooling, the per-GPU share of the DDP job
            #   large_5s_b32  configs[3]: wav2vec2-large geometry (24 layers, H = 1024), 5 s clips, 32 utterances per GPU
            #   ecapa_f32_b66 configs[4]: ECAPA-TDNN, with its per-family HBM roofline (tools/ecapa_bench.py)
            del trainer, plan, store
            torch.cuda.empty_cache()

            def leg(model, pooling, seconds, batch, adt, note):
                c = W2V2Config.from_huggingface_id("facebook/wav2vec2-" + model)
                ns = int(round(seconds * 16000))
                st2 = ParamStore(c, dev, adt, head="aam", num_speakers=args.speakers, freeze_cnn=True,
                                 attentive_pool=pooling == "attentive", embed_dim=c.hidden_size * 2)
                st2.init_weights(seed=20211)
                pl2 = Plan(st2, batch, ns, train=True, reg=reg, seed=7, pooling=pooling)
                tr2 = SpeakerTrainer(st2, pl2, OneCycle(max_lr=5e-5, total_steps=max(total, 10)), layerdrop_seed=1234, mask_seed=7)
                w2, l2 = synth_batch(batch, ns, args.speakers, seed=42133724, device=dev)
                # (LayerDrop draws decide how many layers a step runs: few steps = a noisy figure -- the leg reports the
                # layers it skipped, and runs the warm-up draws of the headline run first so both see the same sequence)
                for _ in range(args.warmup):
                    tr2.train_step(w2, l2)
                torch.cuda.synchronize()
                # equal footing with the headline: the same number of timed steps behind the same warm-up and the same
                # LayerDrop seed -> a 12-layer leg sees the headline's own skip sequence; every leg also reports the
                # TFLOP/s of the work it actually did (LayerDrop-adjusted), which is comparable whatever was drawn
                nst, nskip = args.steps, 0
                t1 = time.perf_counter()
                for _ in range(nst):
                    ls, _ = tr2.train_step(w2, l2)
                    nskip += len(pl2._skip)
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t1) / nst
                flz = c.flops_per_utt(ns, args.speakers)
                fl2 = flz["train_frozen_cnn"]
                res = {"ms_per_step": round(1e3 * dt2, 3), "value": round(batch / dt2, 2), "unit": "utterances/sec",
                       "steps": nst, "warmup": args.warmup, "layerdrop_skipped_layers_per_step": round(nskip / nst, 2),
                       "model_tflops": round(fl2 * batch / dt2 / 1e12, 1),
                       "model_tflops_layerdrop_adjusted": round((fl2 - 3.0 * flz["layer"] * nskip / nst) * batch / dt2 / 1e12, 1),
                       "final_loss": round(float(ls), 4),
                       "workload": f"wav2vec2-{model} + AAM-softmax({args.speakers}), {pooling} pooling, {seconds:g} s "
                                   f"synthetic audio, bs={batch}, {str(adt).split('.')[-1]}, fwd+bwd+Adam -- {note}"}
                del tr2, pl2, st2
                torch.cuda.empty_cache()
                return res
            also = {}
            also["bf16"] = leg("base", "mean+std", args.seconds, args.batch, torch.bfloat16,
                               "BASELINE configs[1] in the bf16 mode (embedding rel-L2 vs the reference 8.4e-3, bound "
                               "3e-2 asserted; fp16: < 1e-3 asserted)")
            also["attentive_b66"] = leg("base", "attentive", 3.0, 66, torch.float16, "BASELINE configs[2], per-GPU share")
            also["large_5s_b32"] = leg("large", "mean+std", 5.0, 32, torch.float16, "BASELINE configs[3], per-GPU share")
            try:
                from types import SimpleNamespace
                from tools.ecapa_bench import bench_ecapa
                e = bench_ecapa(SimpleNamespace(batch=66, frames=300, steps=5, warmup=2, dtype="f32"), 1, 0, dev, None,
                                emit=False)
                also["ecapa_f32_b66"] = {k: e[k] for k in ("ms_per_step", "value", "unit", "steps", "warmup", "dtype", "mfma",
                                                          "model_tflops_per_gpu", "roofline", "roofline_families",
                                                          "gemm_mfma") if k in e}
                also["ecapa_f32_b66"]["workload"] = e["config"]["workload"]
            except Exception as ex:                    # a side leg must never cost the headline line
                also["ecapa_f32_b66"] = {"error": repr(ex)}
            out["also"] = also
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
