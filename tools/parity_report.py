#!/usr/bin/env python3
"""Measured parity numbers of the HIP path against the REFERENCE-generated goldens (tests/golden/*.npz, produced by
tests/golden/make_goldens.py from the imported reference), per golden x arithmetic mode -- the numbers behind the asserts
of tests/test_parity_gpu.py, kept as a file (VERDICT r3 "keep the parity numbers").

    python3 tools/parity_report.py [out.json]         (GPU box; tools/round_final.sh writes gpurun_out/<tag>_parity.json)

Per entry: embedding rel-L2 over the batch and per utterance (max / median), loss error, worst relative gradient error
(full gradients for the tiny golden, gradient norms for the base golden).  Nothing here reads /root/reference."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import GOLDEN, rel_l2
from oracle import w2v2_oracle as O            # (checker side: seeded weights / inputs of the goldens)
from test_parity_gpu import _cfgs, _gscale, _no_reg, _store, load, T

DEV = "cuda"
MODES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def per_utt(e, ref):
    e, ref = e.double().cpu(), T(ref).double()
    r = (e - ref).norm(dim=1) / ref.norm(dim=1)
    return {"batch_rel_l2": float((e - ref).norm() / ref.norm()), "per_utterance_max": float(r.max()),
            "per_utterance_median": float(r.median()), "per_utterance_p90": float(r.quantile(0.9)), "utterances": int(r.numel())}


def tiny(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g1_tiny.npz")
    cfg, ocfg = _cfgs("tiny")
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0
    wav, label, mask = T(g["wav"]).to(DEV), T(g["label"]).to(DEV), T(g["mask"])
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=_no_reg())
    st.zero_grad()
    emb = plan.embed(wav, mask.to(DEV))
    loss, _ = plan.head_forward_backward(label)
    plan.backward()
    torch.cuda.synchronize()
    gs = _gscale(st)
    worst, worst_name, gmax = 0.0, "", max(float(np.linalg.norm(g["grad." + (n[len("wav2vec.model."):] if n.startswith("wav2vec.model.") else n)]))
                                         for n in st.shapes if st.is_trainable(n))
    for n in st.shapes:
        if not st.is_trainable(n):
            continue
        ref = g["grad." + (n[len("wav2vec.model."):] if n.startswith("wav2vec.model.") else n)].astype(np.float64)
        if np.linalg.norm(ref) < 1e-4 * gmax:          # analytically-zero gradients (k_proj bias): rounding noise
            continue
        err = float(np.linalg.norm(st.g(n).cpu().numpy().astype(np.float64) / gs - ref) / np.linalg.norm(ref))
        if err > worst:
            worst, worst_name = err, n
    out = {"train_embedding": per_utt(emb, g["embedding"]), "loss_rel_err": abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])),
           "worst_gradient_rel_l2": worst, "worst_gradient": worst_name}
    ev = Plan(st, 2, wav.shape[-1], train=False)
    out["eval_embedding"] = per_utt(ev.embed(wav), g["eval.mean+std"])
    return out


def base(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g2_base.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, "aam", 5994)
    if dtype == torch.float16:
        st.scaler[0] = 1024.0
    wav, label = O.synth_batch(2, 48000, 5994, seed=42133724)
    wav, label = wav.to(DEV), label.to(DEV)
    ev = Plan(st, 2, 48000, train=False)
    out = {"eval_embedding": per_utt(ev.embed(wav), g["eval.mean+std"])}
    del ev
    tr = Plan(st, 2, 48000, train=True, reg=_no_reg())
    st.zero_grad()
    emb = tr.embed(wav, T(g["mask"]).to(DEV))
    loss, _ = tr.head_forward_backward(label)
    tr.backward()
    torch.cuda.synchronize()
    out["train_embedding"] = per_utt(emb, g["train.embedding"])
    out["loss_rel_err"] = abs(float(loss) - float(g["train.loss"])) / abs(float(g["train.loss"]))
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    gs, gmax = _gscale(st), max(norms.values())
    worst, worst_name = 0.0, ""
    for n, ref in norms.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if not st.is_trainable(name) or ref < 1e-3 * gmax:
            continue
        err = abs(float(st.g(name).double().norm()) / gs - ref) / ref
        if err > worst:
            worst, worst_name = err, n
    out["worst_gradient_norm_rel_err"], out["worst_gradient"] = worst, worst_name
    return out


def base2(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g10_base2.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1, seed=777)
    wav, _ = O.synth_batch(8, 80000, 5994, seed=31337)
    ev = Plan(st, 8, 80000, train=False)
    return {"eval_embedding": per_utt(ev.embed(wav.to(DEV)), g["eval.mean+std"])}


def base66(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g11_base66.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    wav, _ = O.synth_batch(66, 48000, 5994, seed=42133724)
    ev = Plan(st, 66, 48000, train=False)
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    out = {"eval_embedding": per_utt(e, g["eval.mean+std"]),
           "hidden_states_sample_rel_l2": rel_l2(ev.out[:, ::32, ::32].float().cpu(), g["eval.last_hidden.sample"])}
    return out


def long_utt(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g13_long.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    wav, _ = O.synth_batch(1, 320000, 5994, seed=90017)
    ev = Plan(st, 1, 320000, train=False)
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    return {"eval_embedding": per_utt(e, g["eval.mean+std"]),
            "hidden_states_sample_rel_l2": rel_l2(ev.out[:, ::37, ::16].float().cpu(), g["eval.last_hidden.sample"])}


def seed3(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g14_seed3.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1, seed=4099)
    wav, _ = O.synth_batch(6, 64000, 5994, seed=60611)
    ev = Plan(st, 6, 64000, train=False)
    return {"eval_embedding": per_utt(ev.embed(wav.to(DEV)), g["eval.mean+std"])}


def eer(dtype):
    from w2v2_speaker_amd.data.synthetic import score_trials, synth_trial_set
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.eval_metrics import calculate_eer, calculate_mdc
    g = load("g12_eer.npz")
    wav, _spk, _keys, trials = synth_trial_set()
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    ev = Plan(st, wav.shape[0], wav.shape[1], train=False)
    e = ev.embed(T(wav).to(DEV))
    torch.cuda.synchronize()
    gt, sc = score_trials(e.float().cpu().numpy(), trials)
    er, _ = calculate_eer(gt, sc)
    md, _ = calculate_mdc(gt, sc)
    return {"eval_embedding": per_utt(e, g["embedding"]), "eer_hip": float(er), "eer_reference": float(g["eer"]),
            "min_dcf_hip": float(md), "min_dcf_reference": float(g["mdc"]),
            "max_abs_score_diff": float(np.abs(np.array(sc) - g["scores"]).max()), "trials": len(trials)}


def main():
    rep = {"what": "HIP path vs reference-generated goldens (tests/golden, make_goldens.py); rel-L2 = ||got - ref|| / ||ref||",
           "device": torch.cuda.get_device_name(0), "goldens": {}}
    for name, fn, desc in (("g1_tiny", tiny, "tiny geometry, B=2: train forward + loss + every gradient, eval embedding"),
                           ("g2_base", base, "w2v2-base, B=2, 3 s: eval / train embedding, AAM loss, gradient norms"),
                           ("g10_base2", base2, "w2v2-base, other seed, B=8, 5 s: eval embedding"),
                           ("g11_base66", base66, "w2v2-base, B=66, 3 s (BASELINE configs[1] at its own size): eval embedding"),
                           ("g12_eer", eer, "w2v2-base, 32 synthetic trial utterances (8 speakers x 4), 496 trials: embedding, "
                                            "scores, EER / minDCF against the reference's own evaluator"),
                           ("g13_long", long_utt, "w2v2-base, ONE 20 s utterance at batch size 1 (T = 999): eval embedding"),
                           ("g14_seed3", seed3, "w2v2-base, third weight seed, B=6, 4 s (T = 199): eval embedding")):
        rep["goldens"][name] = {"description": desc}
        for mode, dt in MODES.items():
            rep["goldens"][name][mode] = fn(dt)
            torch.cuda.empty_cache()
            e = rep["goldens"][name][mode]["eval_embedding"]
            print(f"{name:11s} {mode:5s} eval embedding rel-L2 {e['batch_rel_l2']:.3e}  per-utterance max {e['per_utterance_max']:.3e}", flush=True)
    txt = json.dumps(rep, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main()
