"""GPU parity tests: the HIP path (engine.Plan through the C ABI) against the CPU oracle and the
reference-generated golden vectors, on identical seeded inputs.

Tolerances (BASELINE.json north_star: "embeddings within 1e-3 rel-L2 of the reference"):
  * f32 parity mode (exact-f32 GEMMs): embeddings rel-L2 < 1e-3 (asserted at 1e-4), gradients < 2e-3.
  * fp16 mode (what bench.py runs by default; the reference's own fp16-AMP arithmetic): operands carry an 11-bit
    significand; the bound is asserted per test below (base model: see test_base_16bit_...), gradients under the
    dynamic loss scale are compared after dividing by it.
  * bf16 mode (8-bit significand, kept as an option): embeddings rel-L2 < 3e-2, gradient norms within 8 %.
The error budget behind these numbers: tests/debug/error_budget*.py (CPU simulation of the storage roundings).
Run with -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import w2v2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfgs(name):
    from w2v2_speaker_amd.config import W2V2Config
    return (W2V2Config.tiny(), O.OracleConfig.tiny()) if name == "tiny" else (W2V2Config(), O.OracleConfig.base())


def _store(cfg, ocfg, dtype, head, C, seed=20211, embed_dim=None):
    from w2v2_speaker_amd.params import ParamStore
    st = ParamStore(cfg, DEV, dtype, head=head, num_speakers=C, embed_dim=embed_dim)
    sd = O.make_state_dict(ocfg, seed)
    E = st.embed_dim
    if head == "aam":
        sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (C, E), seed)
    elif head == "ce":
        sd["fc_list.0.0.weight"] = O.synth_tensor("fc_list.0.0.weight", (C, E), seed)
        sd["fc_list.0.0.bias"] = O.synth_tensor("fc_list.0.0.bias", (C,), seed)
    st.load_state_dict(sd)
    return st, sd


def _no_reg():
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    return Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                                        hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)


def _gscale(st) -> float:
    """Loss scale carried by every gradient of an fp16 store (1 otherwise)."""
    return float(st.scaler[0]) if st.scaler is not None else 1.0


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_tiny_all_stages_loss_and_every_gradient_vs_reference_golden(dtype):
    """Golden G1 was produced by the reference (HF model via the reference wrapper + its pooling and
    AAM modules): stage activations, loss, softmax and the gradient of every parameter."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g1_tiny.npz")
    cfg, ocfg = _cfgs("tiny")
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0      # B = 2: d loss / d cos of the label column is 33x the workload's (B = 66)
    wav, label, mask = T(g["wav"]).to(DEV), T(g["label"]).to(DEV), T(g["mask"])
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=_no_reg())
    st.zero_grad()
    emb = plan.embed(wav, mask.to(DEV))
    loss, sm = plan.head_forward_backward(label)
    plan.backward()
    torch.cuda.synchronize()
    f32, f16 = dtype == torch.float32, dtype == torch.float16
    # measured (round 2): fp16 stages 1.0-1.3e-3, embedding 8.7e-4, worst gradient 8.6e-4; bf16 8x those
    tol = 1e-4 if f32 else (3e-3 if f16 else 3e-2)
    B, Tn, H = plan.out.shape
    assert rel_l2(plan.conv[-1].float().cpu(), g["stage.conv_out"]) < (1e-5 if f32 else (2e-3 if f16 else 1e-2))
    assert rel_l2(plan.X[0].float().cpu().view(B, Tn, H), g["stage.enc_in"]) < tol
    for l in range(cfg.num_hidden_layers):
        assert rel_l2(plan.X[l + 1].float().cpu().view(B, Tn, H), g[f"stage.layer{l}"]) < tol, l
    assert rel_l2(emb.cpu(), g["embedding"]) < (1e-4 if f32 else (2e-3 if f16 else 3e-2))
    assert abs(float(loss) - float(g["loss"])) < (1e-4 if f32 else (2e-3 if f16 else 5e-2)) * abs(float(g["loss"]))
    assert rel_l2(sm.cpu(), g["softmax"]) < (1e-3 if f32 else (1e-2 if f16 else 0.15))
    gtol = 2e-3 if f32 else (6e-3 if f16 else 0.12)
    gs = _gscale(st)
    worst = 0.0
    for name in st.shapes:
        if not st.is_trainable(name):
            continue
        key = "grad." + (name[len("wav2vec.model."):] if name.startswith("wav2vec.model.") else name)
        ref = g[key]
        got = st.g(name).cpu().numpy().astype(np.float64) / gs
        err = np.linalg.norm(got.astype(np.float64) - ref)
        floor = 1e-6 if f32 else (2e-4 if f16 else 2e-3)
        assert err <= gtol * np.linalg.norm(ref) + floor, (name, err, np.linalg.norm(ref))
        worst = max(worst, err / (np.linalg.norm(ref) + 1e-12))
    print("worst grad rel err", worst)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_pre_ln_layer_norm_conv_family_vs_reference_golden(dtype):
    """VERDICT r5 item 6 / SURVEY App. A.12: the "-lv60" / xlsr checkpoint family -- pre-LN encoder (HF:611-654,729-802), a
    LayerNorm after every convolution (HF:275-299), convolutions with bias -- through the HIP engine against
    tests/golden/g19_tiny_stable.npz (the reference wrapper with those HF config flags, tiny geometry, three blocks): every
    stage, the loss and the gradient of every trainable parameter, without and with a LayerDrop skip of the middle block."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    g = load("g19_tiny_stable.npz")
    kw = dict(num_hidden_layers=3, do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True)
    cfg, ocfg = dataclasses.replace(W2V2Config.tiny(), **kw), dataclasses.replace(O.OracleConfig.tiny(), **kw)
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    assert "wav2vec.model.feature_extractor.conv_layers.4.conv.bias" in st.shapes
    assert "wav2vec.model.feature_extractor.conv_layers.6.layer_norm.weight" in st.shapes
    if st.scaler is not None:
        st.scaler[0] = 256.0
    wav, label, mask = T(g["wav"]).to(DEV), T(g["label"]).to(DEV), T(g["mask"])
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=_no_reg())
    assert plan.stable and plan.ln_conv
    f32, f16 = dtype == torch.float32, dtype == torch.float16
    tol = 1e-4 if f32 else (3e-3 if f16 else 3e-2)
    gtol = 2e-3 if f32 else (8e-3 if f16 else 0.12)
    for tag, skip in (("", ()), ("skip1.", (1,))):
        st.zero_grad()
        emb = plan.embed(wav, mask.to(DEV), skip)
        loss, _ = plan.head_forward_backward(label)
        plan.backward()
        torch.cuda.synchronize()
        B, Tn, H = plan.out.shape
        if not skip:
            assert rel_l2(plan.conv[-1].float().cpu(), g["stage.conv_out"]) < (1e-5 if f32 else (2e-3 if f16 else 1.5e-2))
            assert rel_l2(plan.pos.float().cpu().view(B, Tn, H), g["stage.enc_in"]) < tol         # x_0 = hx + pos
            for l in range(cfg.num_hidden_layers):      # the un-normalised residual stream after block l
                assert rel_l2(plan.lb[l].f.float().cpu().view(B, Tn, H), g[f"stage.layer{l}"]) < tol, l
        assert rel_l2(plan.out.float().cpu(), g[tag + "last_hidden"]) < tol
        assert rel_l2(emb.cpu(), g[tag + "embedding"]) < (1e-4 if f32 else (2e-3 if f16 else 3e-2))
        assert abs(float(loss) - float(g[tag + "loss"])) < (1e-4 if f32 else (3e-3 if f16 else 5e-2)) * abs(float(g[tag + "loss"]))
        gs = _gscale(st)
        worst = 0.0
        for name in st.shapes:
            if not st.is_trainable(name):
                continue
            key = tag + "grad." + (name[len("wav2vec.model."):] if name.startswith("wav2vec.model.") else name)
            ref = g[key]
            got = st.g(name).cpu().numpy().astype(np.float64) / gs
            err = np.linalg.norm(got - ref)
            floor = 1e-6 if f32 else (3e-4 if f16 else 3e-3)
            assert err <= gtol * np.linalg.norm(ref) + floor, (tag, name, err, np.linalg.norm(ref))
            worst = max(worst, err / (np.linalg.norm(ref) + 1e-12))
        print(f"pre-LN family {dtype} {tag or 'full'}: worst gradient rel err {worst:.3e}")
    # eval plan (single buffer set, ping-pong X)
    ev = Plan(st, 2, wav.shape[-1], train=False)
    e = ev.embed(wav)
    torch.cuda.synchronize()
    assert rel_l2(e.cpu(), g["eval.mean+std"]) < (1e-4 if f32 else (2e-3 if f16 else 3e-2))
    with pytest.raises(NotImplementedError):
        Plan(st, 2, wav.shape[-1], train=False, insert_cls_token=True, pooling="first+cls")


def test_large_lv60_geometry_cut_vs_oracle():
    """`W2V2Config.from_huggingface_id("facebook/wav2vec2-large-lv60")` (H = 1024, 16 heads, FFN 4096, pre-LN, layer-norm
    convolutions with bias) cut to 2 blocks, 2 s clips: eval embedding, train-mode loss and every trainable gradient norm of
    the HIP path (exact-f32 mode) against the oracle restatement that g19_tiny_stable pins; eval embedding in fp16 / bf16."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    cfg = dataclasses.replace(W2V2Config.from_huggingface_id("facebook/wav2vec2-large-lv60"), num_hidden_layers=2)
    ocfg = dataclasses.replace(O.OracleConfig.large(), num_hidden_layers=2, do_stable_layer_norm=True,
                               feat_extract_norm="layer", conv_bias=True)
    assert cfg.do_stable_layer_norm and cfg.feat_extract_norm == "layer" and cfg.conv_bias and cfg.hidden_size == 1024
    B, N, C = 2, 32000, 101
    wav, label = O.synth_batch(B, N, C, seed=78)
    st, sd = _store(cfg, ocfg, torch.float32, "aam", C)
    sdg = {k: v.clone().requires_grad_(k.startswith("encoder") or k.startswith("feature_projection")
                                       or k in ("masked_spec_embed", "loss_fn.fc_weights")) for k, v in sd.items()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    emb_ref = O.speaker_embedding(wav, sdg, ocfg)
    loss_ref, _ = O.aam_softmax(emb_ref, sdg["loss_fn.fc_weights"], label)
    loss_ref.backward()
    tr = Plan(st, B, N, train=True, reg=_no_reg())
    st.zero_grad()
    emb = tr.embed(wav.to(DEV))
    loss, _ = tr.head_forward_backward(label.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    assert rel_l2(emb.cpu(), emb_ref.detach()) < 1e-4
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    gmax = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    checked = 0
    for n, v in sdg.items():
        if v.grad is None:
            continue
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if not st.is_trainable(name):
            continue
        ref, got = float(v.grad.double().norm()), float(st.g(name).double().norm())
        assert abs(got - ref) <= 2e-3 * ref + 1e-6 * gmax, (n, got, ref)
        checked += 1
    assert checked > 30
    del tr
    for lp, bound in ((torch.float16, 4e-3), (torch.bfloat16, 3e-2)):
        stb, _ = _store(cfg, ocfg, lp, "aam", C)
        evb = Plan(stb, B, N, train=False)
        eb = evb.embed(wav.to(DEV))
        torch.cuda.synchronize()
        err = rel_l2(eb.cpu(), emb_ref.detach())
        print(f"large-lv60 geometry (2-block cut) {lp}: embedding rel-L2 vs the oracle {err:.3e}")
        assert err < bound, (lp, err)
        del evb, stb


def test_layer_norm_convolution_stack_at_base_width_vs_oracle():
    """The layer-norm convolution family at the real width (512 channels, k = 10 / 3 / 2, biases): conv_features() of the
    HIP path against the (g19-pinned) oracle restatement on one second of audio, f32 and fp16."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    kw = dict(num_hidden_layers=1, do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True)
    cfg, ocfg = dataclasses.replace(W2V2Config(), **kw), dataclasses.replace(O.OracleConfig.base(), **kw)
    wav, _ = O.synth_batch(3, 16000, 10, seed=5)
    sd = O.make_state_dict(ocfg, 31)
    with torch.no_grad():
        ref = O.feature_extractor(wav[:, 0], sd, ocfg).transpose(1, 2)
    for dtype, bound in ((torch.float32, 1e-5), (torch.float16, 2e-3)):
        st = ParamStore(cfg, DEV, dtype, head=None, num_speakers=1)
        st.load_state_dict(sd)
        ev = Plan(st, 3, 16000, train=False)
        out = ev.conv_features(wav.to(DEV))
        torch.cuda.synchronize()
        err = rel_l2(out.float().cpu(), ref)
        print(f"layer-norm conv stack {dtype}: rel-L2 {err:.3e}")
        assert err < bound, (dtype, err)
        del ev, st


def test_tiny_ce_head_and_other_pools_vs_golden():
    from w2v2_speaker_amd.engine import Plan
    g = load("g1_tiny.npz")
    cfg, ocfg = _cfgs("tiny")
    wav = T(g["wav"]).to(DEV)
    st, sd = _store(cfg, ocfg, torch.float32, "ce", 10)
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=_no_reg())
    st.zero_grad()
    plan.embed(wav, T(g["mask"]).to(DEV))
    loss, sm = plan.head_forward_backward(T(g["label"]).to(DEV))
    plan.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["ce.loss"])) < 1e-4
    assert rel_l2(sm.cpu(), g["ce.softmax"]) < 1e-4
    # CE-head gradients against oracle autograd
    emb = T(g["embedding"]).clone().requires_grad_(True)
    W = sd["fc_list.0.0.weight"].clone().requires_grad_(True)
    b = sd["fc_list.0.0.bias"].clone().requires_grad_(True)
    l2, _ = O.ce_head(emb, W, b, T(g["label"]))
    l2.backward()
    assert rel_l2(st.g("fc_list.0.0.weight").cpu(), W.grad) < 1e-3
    assert rel_l2(st.g("fc_list.0.0.bias").cpu(), b.grad) < 1e-3
    assert rel_l2(plan.demb.cpu(), emb.grad) < 1e-3
    for pool in ("mean+std", "mean", "max", "first", "middle", "last", "quantile"):
        st2, _ = _store(cfg, ocfg, torch.float32, None, 10)
        p = Plan(st2, 2, wav.shape[-1], train=False, pooling=pool)
        e = p.embed(wav)
        torch.cuda.synchronize()
        assert rel_l2(e.cpu(), g["eval." + pool]) < 1e-4, pool
    p = Plan(st2, 2, wav.shape[-1], train=False, pooling="first+cls", insert_cls_token=True)
    e = p.embed(wav)
    torch.cuda.synchronize()
    assert rel_l2(p.out.float().cpu(), g["eval.cls.last_hidden"]) < 1e-4
    assert rel_l2(e.cpu(), g["eval.first+cls"]) < 1e-4


def test_base_f32_embeddings_within_1e3_of_reference_and_grad_norms():
    """BASELINE north_star: embeddings within 1e-3 rel-L2 of the reference (f32 parity mode)."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g2_base.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, torch.float32, "aam", 5994)
    wav, label = O.synth_batch(2, 48000, 5994, seed=42133724)
    wav, label = wav.to(DEV), label.to(DEV)
    ev = Plan(st, 2, 48000, train=False)
    e = ev.embed(wav)
    torch.cuda.synchronize()
    err = rel_l2(e.cpu(), g["eval.mean+std"])
    print("base f32 eval embedding rel-L2 vs reference:", err)
    assert err < 1e-4
    assert rel_l2(ev.out.float().cpu()[:, ::16, ::16], g["eval.last_hidden.sample"]) < 1e-4
    cl = Plan(st, 2, 48000, train=False, pooling="first+cls", insert_cls_token=True)
    assert rel_l2(cl.embed(wav).cpu(), g["eval.first+cls"]) < 1e-4
    del ev, cl
    tr = Plan(st, 2, 48000, train=True, reg=_no_reg())
    st.zero_grad()
    emb = tr.embed(wav, T(g["mask"]).to(DEV))
    loss, sm = tr.head_forward_backward(label)
    tr.backward()
    torch.cuda.synchronize()
    assert rel_l2(emb.cpu(), g["train.embedding"]) < 1e-4
    assert abs(float(loss) - float(g["train.loss"])) < 1e-4 * abs(float(g["train.loss"]))
    assert rel_l2(sm.cpu()[:, ::37], g["train.softmax.sample"]) < 1e-3
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    floor = 1e-6 * max(norms.values())
    for n, ref in norms.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if not st.is_trainable(name):
            continue
        got = st.g(name)
        assert abs(float(got.double().norm()) - ref) <= 2e-3 * ref + floor, (n, float(got.double().norm()), ref)
        head = got.flatten()[:32].cpu().numpy()
        assert np.allclose(head, g["gradhead." + n], rtol=5e-3, atol=2e-3 * ref / np.sqrt(got.numel()) + floor), n


# embedding rel-L2 of the base model against the reference golden, per 16-bit format.  fp16 is the mode bench.py
# runs and carries north_star's bar (< 1e-3): measured 7.7e-4 (eval) / 7.6e-4 (train forward).  Plain fp16 operands
# give 1.04e-3, exactly what the CPU simulation of the storage roundings predicts (tests/debug/error_budget2.py):
# 0.84e-3 of it is the rounding of the encoder WEIGHTS, a systematic, token-independent error that mean pooling cannot
# average out, and most of that comes from the value / output projections -- hence their two-term weights
# (ParamStore.two_term, w2v2_gemm_desc.k_ext), which bring the simulation to 7.5e-4.
EMB_BOUND = {torch.float16: 1e-3, torch.bfloat16: 3e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_base_second_golden_other_seed_batch8_5s(dtype):
    """A second, independent reference golden for the embedding bound (tests/golden/g10_base2.npz, make_goldens.py
    `base2`): other weights (seed 777), other utterances, B = 8, 5 s clips (T = 249).  The 1e-3 rel-L2 target of the
    benchmarked fp16 mode must hold here too -- the two-term-weight choice was tuned on g2_base only."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g10_base2.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1, seed=777)
    wav, _ = O.synth_batch(8, 80000, 5994, seed=31337)
    ev = Plan(st, 8, 80000, train=False)
    assert ev.T == 249
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    err = rel_l2(e.cpu(), g["eval.mean+std"])
    hs = rel_l2(ev.out[:, ::16, ::16].float().cpu(), g["eval.last_hidden.sample"])
    print(f"second base golden {dtype}: embedding rel-L2 {err:.3e}, sampled hidden states {hs:.3e}")
    assert err < {torch.float32: 1e-4, **EMB_BOUND}[dtype], err
    per_utt = (e.cpu() - T(g["eval.mean+std"])).norm(dim=1) / T(g["eval.mean+std"]).norm(dim=1)
    # every utterance, not the mean (measured, profiles/r04_parity.json: fp16 worst utterance 9.3e-4 here, 8.5e-4 of 66 at B = 66)
    assert float(per_utt.max()) < {torch.float32: 1e-4, **EMB_BOUND}[dtype], per_utt


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_base_b66_embeddings_vs_reference_golden_at_benchmark_size(dtype):
    """BASELINE configs[1] compared with the REFERENCE at its own size: the 66 utterances bench.py
    steps on, 3 s each, eval embeddings of the reference's wrapper + mean+std pooling (tests/golden/g11_base66.npz,
    make_goldens.py `base66`).  Bounds: f32 1e-4; fp16 (the benchmarked mode) 1e-3 over the batch AND for every single
    utterance; bf16 3e-2.  The measured values are kept in profiles/r04_parity.json (tools/parity_report.py)."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g11_base66.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    wav, _ = O.synth_batch(66, 48000, 5994, seed=42133724)
    ev = Plan(st, 66, 48000, train=False)
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    ref = T(g["eval.mean+std"])
    bound = {torch.float32: 1e-4, **EMB_BOUND}[dtype]
    err = rel_l2(e.cpu(), ref)
    per_utt = (e.cpu() - ref).norm(dim=1) / ref.norm(dim=1)
    print(f"B=66 golden {dtype}: embedding rel-L2 {err:.3e}, per utterance max {float(per_utt.max()):.3e} "
          f"median {float(per_utt.median()):.3e}")
    assert err < bound, err
    assert float(per_utt.max()) < bound, per_utt
    assert rel_l2(ev.out[:, ::32, ::32].float().cpu(), g["eval.last_hidden.sample"]) < {torch.float32: 1e-4, torch.float16: 3e-3,
                                                                                       torch.bfloat16: 3e-2}[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_eer_on_the_synthetic_trial_set_hip_vs_reference(dtype):
    """The "eval EER" half of the metric at the benchmarked size and precision (VERDICT r4 missing 3): the 32 utterances of
    w2v2_speaker_amd.data.synthetic.synth_trial_set through the HIP engine (eval, mean+std), scored on all 496 pairs like
    the reference's evaluator (ref: speaker_recognition_evaluator.py:46-115), against tests/golden/g12_eer.npz = the
    REFERENCE's embeddings / scores / EER for the same waveforms (reference EER 0.104: target and non-target scores
    overlap, so the figure reacts to embedding errors).  Bounds: embeddings as everywhere (f32 1e-4, fp16 1e-3 per
    utterance); every trial score within 1e-6 (f32) / 5e-5 (fp16: scores live in 0.998..1, the target / non-target gap is
    7e-4); EER equal (f32) / within one target trial = 1/48 (fp16), minDCF within 0.05."""
    from w2v2_speaker_amd.data.synthetic import score_trials, synth_trial_set
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.eval_metrics import calculate_eer, calculate_mdc
    g = load("g12_eer.npz")
    wav, spk, keys, trials = synth_trial_set()
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    ev = Plan(st, wav.shape[0], wav.shape[1], train=False)
    e = ev.embed(T(wav).to(DEV)).cpu()
    torch.cuda.synchronize()
    ref = T(g["embedding"])
    per_utt = (e - ref).norm(dim=1) / ref.norm(dim=1)
    f32 = dtype == torch.float32
    assert float(per_utt.max()) < (1e-4 if f32 else 1e-3), per_utt
    gt, sc = score_trials(e.numpy(), trials)
    dsc = float(np.abs(np.array(sc) - g["scores"]).max())
    eer, _ = calculate_eer(gt, sc)
    mdc, _ = calculate_mdc(gt, sc)
    print(f"EER trial set {dtype}: hip {eer:.5f} reference {float(g['eer']):.5f}; minDCF {mdc:.4f} vs {float(g['mdc']):.4f}; "
          f"max |score diff| {dsc:.2e}; embedding per-utterance max {float(per_utt.max()):.2e}")
    assert dsc < (1e-6 if f32 else 5e-5), dsc
    assert abs(eer - float(g["eer"])) <= (1e-6 if f32 else 1.0 / 48 + 1e-6), (eer, float(g["eer"]))
    assert abs(mdc - float(g["mdc"])) <= (1e-6 if f32 else 0.05), (mdc, float(g["mdc"]))
    # the centred (z-scored) + length-normed branch of the evaluator (ref: cosine_distance.py:117-127): the direction all
    # embeddings share is removed, so an embedding error of 1e-3 is ~15x larger relative to what is scored; fitted on the
    # HIP embeddings as a reference run would fit on its own.  Scores live in 0.35..0.75 here (reference EER 0.0045).
    from w2v2_speaker_amd.evaluation.speaker.cosine_distance import CosineDistanceEvaluator, EmbeddingSample, EvaluationPair
    smp = [EmbeddingSample(k, v) for k, v in zip(keys, e)]
    cev = CosineDistanceEvaluator(True, True, len(smp))
    cev.fit_parameters([v for v in e], [])
    csc = np.clip((np.array(cev._compute_prediction_scores([(smp[i], smp[j]) for _, i, j in trials])) + 1) / 2, 0, 1)
    cres = cev.evaluate([EvaluationPair(bool(s), keys[i], keys[j]) for s, i, j in trials], smp)
    cd = float(np.abs(csc - g["scores_cl"]).max())
    print(f"  centred + length norm: hip EER {cres['eer']:.5f} reference {float(g['eer_cl']):.5f}; minDCF {cres['mdc']:.4f} vs "
          f"{float(g['mdc_cl']):.4f}; max |score diff| {cd:.2e}")
    if dtype != torch.bfloat16:
        assert cd < (2e-5 if f32 else 5e-3), cd
        assert abs(cres["eer"] - float(g["eer_cl"])) <= (1e-6 if f32 else 1.0 / 48 + 1e-6)
        assert abs(cres["mdc"] - float(g["mdc_cl"])) <= (1e-6 if f32 else 0.05)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_evaluation_length_utterance_and_third_weight_seed_vs_reference(dtype):
    """Two more reference goldens for the embedding bound (the fp16 margin under 1e-3 is a few percent):
    g13_long = ONE 20 s utterance at batch size 1 (T = 999; the reference tests on whole utterances, ref src/main.py:506-514)
    and g14_seed3 = a third weight seed (4099), 6 x 4 s (T = 199).  fp16 < 1e-3 for every utterance."""
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("base")
    bound = {torch.float32: 1e-4, **EMB_BOUND}[dtype]
    hb = {torch.float32: 1e-4, torch.float16: 3e-3, torch.bfloat16: 3e-2}[dtype]
    g = load("g13_long.npz")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    wav, _ = O.synth_batch(1, 320000, 5994, seed=90017)
    ev = Plan(st, 1, 320000, train=False)
    assert ev.T == 999
    e = ev.embed(wav.to(DEV)).cpu()
    err_long = rel_l2(e, g["eval.mean+std"])
    assert rel_l2(ev.out[:, ::37, ::16].float().cpu(), g["eval.last_hidden.sample"]) < hb
    del ev, st
    g = load("g14_seed3.npz")
    st, _ = _store(cfg, ocfg, dtype, None, 1, seed=4099)
    wav, _ = O.synth_batch(6, 64000, 5994, seed=60611)
    ev = Plan(st, 6, 64000, train=False)
    assert ev.T == 199
    e = ev.embed(wav.to(DEV)).cpu()
    ref = T(g["eval.mean+std"])
    per_utt = (e - ref).norm(dim=1) / ref.norm(dim=1)
    print(f"long utterance {dtype}: rel-L2 {err_long:.3e}; third seed: per utterance max {float(per_utt.max()):.3e} "
          f"batch {rel_l2(e, ref):.3e}")
    assert err_long < bound, err_long
    assert float(per_utt.max()) < bound, per_utt
    assert rel_l2(ev.out[:, ::16, ::16].float().cpu(), g["eval.last_hidden.sample"]) < hb


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_five_more_weight_seeds_per_utterance_bound(dtype):
    """How wide the fp16 margin is across WEIGHT seeds: five more reference goldens
    (tests/golden/g16_seeds.npz, make_goldens.py `seeds`: seeds 101 .. 505, 4 utterances of 3 s each), every utterance
    inside the 1e-3 bound of the benchmarked fp16 mode (f32 mode: 1e-4).  The measured maximum is printed."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g16_seeds.npz")
    cfg, ocfg = _cfgs("base")
    worst = 0.0
    for sd_ in g["seeds"].tolist():
        st, _ = _store(cfg, ocfg, dtype, None, 1, seed=sd_)
        wav, _ = O.synth_batch(4, 48000, 5994, seed=7000 + sd_)
        ev = Plan(st, 4, 48000, train=False)
        e = ev.embed(wav.to(DEV)).cpu()
        ref = T(g[f"eval.mean+std.{sd_}"])
        per_utt = (e - ref).norm(dim=1) / ref.norm(dim=1)
        worst = max(worst, float(per_utt.max()))
        assert float(per_utt.max()) < (1e-4 if dtype == torch.float32 else 1e-3), (sd_, per_utt)
        del ev, st
    print(f"five more weight seeds {dtype}: worst utterance rel-L2 {worst:.3e}")


# measured on MI355X (profiles/r06_parity.json): the heavy-tailed family is where the 16-bit modes are weakest -- the bound
# below is what the HIP path is HELD to there; it is reported, not claimed to be 1e-3
OUTLIER_BOUND = {torch.float32: 1e-4, torch.float16: 4e-3, torch.bfloat16: 6e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_heavy_tailed_weight_family_vs_reference_golden(dtype):
    """every golden up to g16 draws its weights from ONE well-conditioned Gaussian family; the pretrained
    model is known for outlier channels.  tests/golden/g17_outlier.npz = the REFERENCE on data/synthetic.py outlier_family
    (six residual channels with every encoder LayerNorm gain x 20, FFN-1 bias entries + 8, convolutions 1 / 3 / 5 x 3: max
    |hidden| / RMS 26-30 in every layer).  The 16-bit modes must stay FINITE and inside OUTLIER_BOUND per utterance; the
    measured figure is printed (and recorded in profiles/r06_parity.json)."""
    from w2v2_speaker_amd.data.synthetic import outlier_family
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    g = load("g17_outlier.npz")
    cfg, ocfg = _cfgs("base")
    st = ParamStore(cfg, DEV, dtype, head=None, num_speakers=1)
    sd = O.make_state_dict(ocfg, 20211)
    sd = {k: T(v) for k, v in outlier_family({k: v.numpy() for k, v in sd.items()}, 20211).items()}
    st.load_state_dict(sd)
    wav, _ = O.synth_batch(4, 48000, 5994, seed=171717)
    ev = Plan(st, 4, 48000, train=False)
    e = ev.embed(wav.to(DEV)).float().cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(e).all() and torch.isfinite(ev.out.float()).all()
    ref = T(g["eval.mean+std"])
    per_utt = (e - ref).norm(dim=1) / ref.norm(dim=1)
    hid = rel_l2(ev.out[:, ::16, ::16].float().cpu(), g["eval.last_hidden.sample"])
    print(f"heavy-tailed family {dtype}: embedding per-utterance max {float(per_utt.max()):.3e}, hidden-state sample rel-L2 {hid:.3e}")
    assert float(per_utt.max()) < OUTLIER_BOUND[dtype], per_utt
    if dtype != torch.float32:
        # context: the REFERENCE's own 16-bit autocast against its own f32 on the same weights and utterances (stored by the
        # golden script: fp16 3.8e-3, bf16 1.5e-2 here; 1.06e-3 fp16 on the Gaussian family, where the HIP path measures 8e-4).
        # The HIP path's mixed precision must be no worse than the reference's, utterance maximum against utterance maximum.
        ref16 = float(g["ref_autocast_fp16.per_utt_err" if dtype == torch.float16 else "ref_autocast_bf16.per_utt_err"].max())
        print(f"   reference autocast {dtype} vs its own f32 on this family: {ref16:.3e}")
        assert float(per_utt.max()) <= 1.05 * ref16, (float(per_utt.max()), ref16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_base_ce_head_1211_vs_reference_golden(dtype):
    """BASELINE configs[0]'s head at its real size: w2v2-base + Linear(1536 -> 1211) + cross-entropy
    (ref: wav2vec2_fc.py:199-210, cross_entropy.py:27-31) on the train-mode embedding of g2_base (injected mask, dropouts
    off): loss and the softmax probability of every label against the reference (`ce.loss`, `ce.softmax.label`)."""
    from w2v2_speaker_amd.engine import Plan
    g = load("g2_base.npz")
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, "ce", 1211)
    wav, label = O.synth_batch(2, 48000, 5994, seed=42133724)
    lab = (label % 1211).to(DEV)
    if st.scaler is not None:
        st.scaler[0] = 256.0
    plan = Plan(st, 2, 48000, train=True, reg=_no_reg())
    st.zero_grad()
    emb = plan.embed(wav.to(DEV), T(g["mask"]).to(DEV))
    loss, sm = plan.head_forward_backward(lab)
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert rel_l2(emb.cpu(), g["train.embedding"]) < (1e-4 if f32 else 1e-3)
    assert abs(float(loss) - float(g["ce.loss"])) < (1e-4 if f32 else 3e-3) * abs(float(g["ce.loss"]))
    got = sm.cpu().gather(1, (label % 1211).view(-1, 1))
    assert rel_l2(got, g["ce.softmax.label"]) < (1e-3 if f32 else 2e-2)



@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_base_16bit_fused_attention_vs_reference_and_vs_unfused(dtype):
    from w2v2_speaker_amd.engine import Plan
    g = load("g2_base.npz")
    cfg, ocfg = _cfgs("base")
    f16 = dtype == torch.float16
    st, _ = _store(cfg, ocfg, dtype, "aam", 5994)
    if f16:
        st.scaler[0] = 1024.0     # B = 2: d loss / d cos of the label column is 66/2 x larger than in the workload
    wav, label = O.synth_batch(2, 48000, 5994, seed=42133724)
    wav, label = wav.to(DEV), label.to(DEV)
    ev = Plan(st, 2, 48000, train=False)
    assert ev.fused
    e = ev.embed(wav).clone()
    un = Plan(st, 2, 48000, train=False, fused_attention=False)
    e2 = un.embed(wav)
    torch.cuda.synchronize()
    err = rel_l2(e.cpu(), g["eval.mean+std"])
    print(f"base {dtype} eval embedding rel-L2 vs reference:", err, " fused vs unfused:", rel_l2(e.cpu(), e2.cpu()))
    assert err < EMB_BOUND[dtype]
    assert rel_l2(e.cpu(), e2.cpu()) < (2e-3 if f16 else 2e-2)
    del ev, un
    tr = Plan(st, 2, 48000, train=True, reg=_no_reg())
    st.zero_grad()
    emb = tr.embed(wav, T(g["mask"]).to(DEV))
    loss, sm = tr.head_forward_backward(label)
    tr.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(st.grad).all()
    assert rel_l2(emb.cpu(), g["train.embedding"]) < EMB_BOUND[dtype]
    assert abs(float(loss) - float(g["train.loss"])) < (1e-3 if f16 else 3e-2) * abs(float(g["train.loss"]))
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    # gradients that cancel analytically (k_proj: softmax is invariant to a shift of the keys) are pure rounding
    # noise of ~1e-3 (bf16) / ~1e-4 (fp16) of the largest gradient norm; the floor has to sit above that noise
    floor = (2e-4 if f16 else 2e-3) * max(norms.values())
    gs = _gscale(st)
    bad = []
    for n, ref in norms.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if not st.is_trainable(name):
            continue
        got = float(st.g(name).double().norm()) / gs
        if abs(got - ref) > (0.01 if f16 else 0.08) * ref + floor:
            bad.append((n, got, ref))
    assert not bad, bad[:10]


def _asp_weights(store, seed=5):
    """Seeded attentive-pooling weights under the speechbrain state-dict names, and as the oracle's dict."""
    from w2v2_speaker_amd.asp import ASP_PREFIX
    sd, od = {}, {}
    for n, shp in store.shapes.items():
        if not n.startswith(ASP_PREFIX):
            continue
        t = O.synth_tensor(n, shp, seed)
        if n.endswith("norm.norm.weight"):
            t = 1.0 + 0.1 * t
        elif n.endswith("conv.conv.weight"):
            t = t * (1.0 / shp[1] ** 0.5)
        sd[n] = t
    od["tdnn.conv.weight"] = sd[ASP_PREFIX + "tdnn.conv.conv.weight"]
    od["tdnn.conv.bias"] = sd[ASP_PREFIX + "tdnn.conv.conv.bias"]
    od["tdnn.norm.weight"] = sd[ASP_PREFIX + "tdnn.norm.norm.weight"]
    od["tdnn.norm.bias"] = sd[ASP_PREFIX + "tdnn.norm.norm.bias"]
    od["conv.weight"] = sd[ASP_PREFIX + "conv.conv.weight"]
    od["conv.bias"] = sd[ASP_PREFIX + "conv.conv.bias"]
    return sd, od


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_attentive_statistics_pooling_step_vs_unpinned_restatement(dtype):
    """SURVEY 8a row a10 / BASELINE configs[2]: wav2vec2 -> attentive statistics pooling (global context, BatchNorm
    with batch statistics) -> AAM.  Embedding, loss and EVERY gradient (encoder + the six pooling tensors) against
    the oracle's autograd.  The oracle restates speechbrain's published definition (speechbrain is not available
    here): parity for this row is unpinned, see oracle.attentive_stat_pool."""
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    cfg, ocfg = _cfgs("tiny")
    B, N, C = 8, 4000, 10
    st = ParamStore(cfg, DEV, dtype, head="aam", num_speakers=C, attentive_pool=True)
    sd = O.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (C, st.embed_dim), 20211)
    asd, aod = _asp_weights(st)
    sd.update(asd)
    st.load_state_dict(sd)
    wav, label = O.synth_batch(B, N, C, seed=11)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.startswith("stat_pooling")}
    aog = {k: v.clone().requires_grad_(True) for k, v in aod.items()}
    hid = O.wav2vec2_forward(wav[:, 0] if wav.dim() == 3 else wav, sdg, ocfg)
    emb_ref = O.attentive_stat_pool(hid, aog)
    loss_ref, _ = O.aam_softmax(emb_ref, sdg["loss_fn.fc_weights"], label)
    loss_ref.backward()
    tr = Plan(st, B, N, train=True, reg=_no_reg(), pooling="attentive")
    if st.scaler is not None:
        st.scaler[0] = 512.0
    gs = _gscale(st)
    st.zero_grad()
    emb = tr.embed(wav.to(DEV))
    loss, _ = tr.head_forward_backward(label.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    f32, f16 = dtype == torch.float32, dtype == torch.float16
    assert rel_l2(emb.cpu(), emb_ref.detach()) < (2e-5 if f32 else 4e-3 if f16 else 3e-2)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < (1e-4 if f32 else 4e-3 if f16 else 3e-2) * abs(float(loss_ref.detach()))
    from w2v2_speaker_amd.asp import ASP_PREFIX
    names = {ASP_PREFIX + "tdnn.conv.conv.weight": aog["tdnn.conv.weight"], ASP_PREFIX + "tdnn.conv.conv.bias": aog["tdnn.conv.bias"],
             ASP_PREFIX + "tdnn.norm.norm.weight": aog["tdnn.norm.weight"], ASP_PREFIX + "tdnn.norm.norm.bias": aog["tdnn.norm.bias"],
             ASP_PREFIX + "conv.conv.weight": aog["conv.weight"], ASP_PREFIX + "conv.conv.bias": aog["conv.bias"]}
    for n, v in sdg.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if v.grad is not None and st.is_trainable(name):
            names[name] = v
    gmax = max(float(v.grad.norm()) for v in names.values())
    tol = 2e-3 if f32 else 0.12
    bad = []
    for name, v in names.items():
        got, ref = st.g(name).double().cpu().reshape(v.grad.shape) / gs, v.grad.double()
        assert torch.isfinite(got).all(), name
        if not f32 and not (name.startswith("loss_fn") or ".conv.conv." in name and "tdnn" not in name):
            # bf16: everything upstream of the BatchNorm backward (a projection orthogonal to {1, rhat} that leaves
            # a small residual of dz here) amplifies rounding noise ~10x in this random tiny model; those stages are
            # checked one by one against f64 on their own bf16 inputs in test_kernels_gpu (attentive pooling stages)
            continue
        err = float((got - ref).norm())
        if err > tol * float(ref.norm()) + (1e-6 if f32 else 2e-3) * gmax:
            bad.append((name, round(err, 4), round(float(ref.norm()), 4)))
    assert not bad, bad
    # eval mode uses the running statistics the training step just updated (BatchNorm1d semantics)
    ev = Plan(st, B, N, train=False, pooling="attentive")
    e2 = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(e2).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_paired_input_bce_step_vs_oracle(dtype):
    """SURVEY 8f row f4 (ref: wav2vec2_paired_input.py:163-207 + binary_cross_entropy.py): two waveforms per pair through
    conv stack + projection, [CLS] left [SEP] right [SEP] through the encoder, Linear(H,1) on token 0, BCE; logits,
    loss and every gradient (incl. the unfrozen conv stack in f32) against the oracle's autograd."""
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    cfg, ocfg = _cfgs("tiny")
    B, N = 3, 4000
    f32 = dtype == torch.float32
    st = ParamStore(cfg, DEV, dtype, head="bce", freeze_cnn=not f32)
    sd = O.make_state_dict(ocfg, 20211)
    sd["linear.weight"] = O.synth_tensor("linear.weight", (1, cfg.hidden_size), 20211)
    sd["linear.bias"] = O.synth_tensor("linear.bias", (1,), 20211)
    st.load_state_dict(sd)
    wl, _ = O.synth_batch(B, N, 2, seed=21)
    wr, _ = O.synth_batch(B, N, 2, seed=22)
    label = torch.tensor([1, 0, 1])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits_ref = O.paired_equality_scores(wl[:, 0], wr[:, 0], sdg, ocfg, sdg["linear.weight"], sdg["linear.bias"])
    loss_ref, pred_ref = O.bce_with_logits(logits_ref, label)
    loss_ref.backward()
    plan = Plan(st, B, N, train=True, reg=_no_reg(), pooling="first", paired=True)
    assert plan.T == 2 * plan.T0 + 3
    if st.scaler is not None:
        st.scaler[0] = 256.0
    gs = _gscale(st)
    st.zero_grad()
    wav = torch.cat([wl[:, 0], wr[:, 0]], dim=0).to(DEV)            # [2B, N]: left utterances, then right
    plan.embed(wav)
    loss, pred = plan.head_forward_backward(label.to(DEV))
    plan.backward()
    torch.cuda.synchronize()
    assert torch.allclose(pred.cpu(), pred_ref, atol=1e-5 if f32 else 3e-2)
    assert abs(float(loss) - float(loss_ref)) < (1e-5 if f32 else 3e-2) * max(1.0, abs(float(loss_ref)))
    gmax = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    bad = []
    for n, v in sdg.items():
        name = n if n.startswith("linear") else "wav2vec.model." + n
        if v.grad is None or not st.is_trainable(name):
            continue
        got, ref = st.g(name).double().cpu().reshape(v.grad.shape) / gs, v.grad.double()
        err = float((got - ref).norm())
        if err > (2e-3 if f32 else 0.12) * float(ref.norm()) + (1e-6 if f32 else 5e-3) * gmax:
            bad.append((n, round(err, 6), round(float(ref.norm()), 6)))
    assert not bad, bad[:8]


def test_large_shape_5s_clips_vs_reference_golden():
    """BASELINE configs[3] geometry (wav2vec2-large: H=1024, 16 heads, FFN 4096; 5 s clips -> T=249, which takes
    the 64-row attention geometry), cut to 2 encoder layers, against tests/golden/g18_large2.npz = the REFERENCE wrapper's
    own "large" branch (ref src/models/wav2vec2.py:115-116; VERDICT r5 item 4b -- this test used to compare with the oracle
    only): eval embedding, train-mode embedding under the injected SpecAugment mask, AAM loss and every trainable gradient
    norm in the exact-f32 mode; eval embedding in the 16-bit modes."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    g = load("g18_large2.npz")
    cfg = dataclasses.replace(W2V2Config.from_huggingface_id("facebook/wav2vec2-large"), num_hidden_layers=2)
    ocfg = dataclasses.replace(O.OracleConfig.large(), num_hidden_layers=2)
    B, N, C = 2, 80000, 211
    assert cfg.num_frames(N) == 249
    wav, label = O.synth_batch(B, N, C, seed=77)
    assert np.array_equal(label.numpy(), g["label"])
    st, _ = _store(cfg, ocfg, torch.float32, "aam", C)
    ev = Plan(st, B, N, train=False)
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(e.cpu(), g["eval.mean+std"]) < 1e-4
    assert rel_l2(ev.out[:, ::16, ::16].float().cpu(), g["eval.last_hidden.sample"]) < 1e-4
    del ev
    tr = Plan(st, B, N, train=True, reg=_no_reg())
    st.zero_grad()
    emb = tr.embed(wav.to(DEV), T(g["mask"]).to(torch.uint8).to(DEV))
    loss, _ = tr.head_forward_backward(label.to(DEV))
    tr.backward()
    torch.cuda.synchronize()
    assert rel_l2(emb.cpu(), g["train.embedding"]) < 1e-4
    assert abs(float(loss) - float(g["train.loss"])) < 1e-4 * abs(float(g["train.loss"]))
    ref = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    gmax = max(ref.values())
    checked = 0
    for n, want in ref.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if name not in st.offsets or not st.is_trainable(name):
            continue
        got = float(st.g(name).double().norm())
        assert abs(got - want) <= 2e-3 * want + 1e-6 * gmax, (n, got, want)
        checked += 1
    assert checked > 30
    del tr
    for lp, bound in ((torch.bfloat16, 3e-2), (torch.float16, 4e-3)):      # fp16 = the benchmarked mode
        stb, _ = _store(cfg, ocfg, lp, "aam", C)
        evb = Plan(stb, B, N, train=False)
        eb = evb.embed(wav.to(DEV))
        torch.cuda.synchronize()
        err = rel_l2(eb.cpu(), g["eval.mean+std"])
        print(f"large (2-layer cut) {lp}: embedding rel-L2 vs the reference {err:.3e}")
        assert err < bound, (lp, err)
        del evb, stb


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_full_batch_no_cross_utterance_mixing_and_determinism(dtype):
    """Size-independent property at the BASELINE size (B = 66, 3 s): an utterance's embedding does not
    depend on its batch neighbours (the reference's own BatchGradientVerification check,
    ref: src/main.py:337-366), and two runs are bit-identical."""
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, dtype, None, 1)
    wav, _ = O.synth_batch(66, 48000, 10, seed=1)
    wav = wav.to(DEV)
    big = Plan(st, 66, 48000, train=False)
    e1 = big.embed(wav).clone()
    e2 = big.embed(wav).clone()
    one = Plan(st, 1, 48000, train=False)
    torch.cuda.synchronize()
    assert torch.equal(e1, e2)
    for i in (0, 17, 65):
        ei = one.embed(wav[i:i + 1]).clone()
        torch.cuda.synchronize()
        # every kernel reduces in a batch-independent order (no atomics on the forward path)
        assert torch.equal(ei, e1[i:i + 1]), (i, rel_l2(ei.cpu(), e1[i:i + 1].cpu()))
    assert torch.isfinite(e1).all()


@pytest.mark.parametrize("pooling", ["mean+std", "attentive"])
def test_b66_fp16_training_steps_of_the_benchmarked_configuration(pooling):
    """BASELINE configs[1] (mean+std) and configs[2] (attentive) exactly as bench.py runs them: w2v2-base, B = 66, 3 s,
    fp16 operands under the dynamic loss scale, dropout / LayerDrop / SpecAugment on, fused Adam.  Size-independent
    properties: two trainers with the same seeds produce BIT-IDENTICAL gradient arenas and parameters (atomic-free
    weight gradients, fixed-order LayerNorm folds; the attentive head's few atomic sums excepted), the loss is
    finite, no step is skipped by the scaler at its default initial scale, the LayerDrop-skipped layers have
    exactly-zero gradients."""
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    cfg, _ = _cfgs("base")
    wav, label = O.synth_batch(66, 48000, 5994, seed=42133724)
    wav, label = wav.to(DEV), label.to(DEV)
    runs = []
    for rep in range(2):
        st = ParamStore(cfg, DEV, torch.float16, head="aam", num_speakers=5994, attentive_pool=pooling == "attentive")
        st.init_weights(seed=20211)
        plan = Plan(st, 66, 48000, train=True, reg=Wav2Vec2RegularisationConfig(), seed=7, pooling=pooling)
        tr = SpeakerTrainer(st, plan, OneCycle(max_lr=5e-5, total_steps=10), layerdrop_seed=77, mask_seed=7)
        buckets = {n: (s, e) for n, s, e in st.grad_buckets()}
        enc = slice(buckets[f"layer{cfg.num_hidden_layers - 1}"][0], buckets["prologue"][0])     # the 12 layer buckets
        losses, skips = [], []
        for i in range(3):
            loss, _ = tr.train_step(wav, label)
            losses.append(float(loss))
            skips.append(tuple(plan._skip))
            if i == 0:
                torch.cuda.synchronize()
                g_first = st.grad[enc].clone()
            for l in plan._skip:
                s_, e_ = buckets[f"layer{l}"]
                assert float(st.grad[s_:e_].abs().max()) == 0.0
        torch.cuda.synchronize()
        assert torch.isfinite(st.grad).all() and torch.isfinite(st.flat).all()
        runs.append((g_first, losses, skips, float(st.scaler[0]), int(st.scaler[3])))
        del tr, plan, st
        torch.cuda.empty_cache()
    (g0, l0, s0, sc0, sk0), (g1, l1, s1, sc1, sk1) = runs
    # step 1 from identical states: the encoder buckets (grouped weight gradients, LayerNorm folds) are bitwise
    # repeatable; the few atomically summed tensors outside them (masked_spec_embed, pos-conv bias) are not, so later
    # steps may differ in the last bits
    assert torch.equal(g0, g1) and float(g0.abs().max()) > 0
    assert all(np.isfinite(l0 + l1)) and l0[0] == l1[0] and s0 == s1, (l0, l1)
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-3
    assert any(len(s) for s in s0), "pick a LayerDrop seed that skips a layer within three steps"
    assert sk0 == 0 and sk1 == 0 and sc0 == 16384.0, "the default initial loss scale must not overflow at B = 66"


def test_train_steps_reduce_loss_and_adam_matches_oracle():
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import Constant
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    cfg, ocfg = _cfgs("tiny")
    st, sd = _store(cfg, ocfg, torch.float32, "aam", 10)
    wav, label = O.synth_batch(4, 4000, 10, seed=3)
    wav, label = wav.to(DEV), label.to(DEV)
    plan = Plan(st, 4, 4000, train=True, reg=_no_reg())
    tr = SpeakerTrainer(st, plan, Constant(1e-3, 0.9))
    p0 = st.flat.clone()
    l0, _ = tr.train_step(wav, label, skip_layers=())
    g0 = st.grad.clone()
    torch.cuda.synchronize()
    # one Adam step == the oracle's restatement of torch.optim.Adam on the same gradient
    p, m, v = p0[:st.n_train].cpu().clone(), torch.zeros(st.n_train), torch.zeros(st.n_train)
    O.adam_step(p, g0.cpu(), m, v, 1, 1e-3, 0.9)
    assert torch.allclose(st.flat[:st.n_train].cpu(), p, atol=1e-6)
    assert torch.equal(st.flat[st.n_train:], p0[st.n_train:])          # frozen CNN untouched
    losses = [float(l0)]
    for _ in range(15):
        l, _ = tr.train_step(wav, label, skip_layers=())
        losses.append(float(l))
    assert losses[-1] < 0.5 * losses[0], losses


def test_regularised_training_step_runs_layerdrop_masks_dropout():
    """Dropout / LayerDrop / SpecAugment on (throughput configuration): finite loss, skipped layers
    get exactly-zero gradients, masked_spec_embed receives gradient."""
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    cfg, ocfg = _cfgs("tiny")
    for dtype in (torch.float32, torch.bfloat16):
        st, _ = _store(cfg, ocfg, dtype, "aam", 10)
        wav, label = O.synth_batch(4, 8000, 10, seed=3)
        plan = Plan(st, 4, 8000, train=True, reg=Wav2Vec2RegularisationConfig(mask_time_length=3))
        tr = SpeakerTrainer(st, plan, OneCycle(1e-3, 10))
        loss, _ = tr.train_step(wav.to(DEV), label.to(DEV), skip_layers=(1,))
        torch.cuda.synchronize()
        assert np.isfinite(float(loss))
        assert float(st.mg("encoder.layers.1.feed_forward.output_dense.weight").abs().max()) == 0.0
        assert float(st.mg("encoder.layers.0.feed_forward.output_dense.weight").abs().max()) > 0.0
        assert float(st.mg("masked_spec_embed").abs().max()) > 0.0
        assert torch.isfinite(st.flat).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_pre_ln_family_regularised_training_steps_and_dropout_backward(dtype):
    """The pre-LN / layer-norm-convolution family under the THROUGHPUT configuration (dropouts, LayerDrop, SpecAugment): three
    trainer steps are finite, LayerDrop-skipped blocks get exactly-zero Linear gradients while their neighbours do not, the
    loss falls on a fixed batch; and (exact-f32 mode) the backward with every dropout site active agrees with central
    differences of the forward along a random direction -- the keep decisions are a pure function of (seed, step, site,
    element), so the loss is a deterministic function of the weights."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    kw = dict(num_hidden_layers=3, do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True)
    cfg, ocfg = dataclasses.replace(W2V2Config.tiny(), **kw), dataclasses.replace(O.OracleConfig.tiny(), **kw)
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0      # B = 4: d loss / d cos is 16x the workload's; the default scale overflows fp16 here
    wav, label = O.synth_batch(4, 8000, 10, seed=3)
    wav, label = wav.to(DEV), label.to(DEV)
    reg = Wav2Vec2RegularisationConfig(mask_time_length=3, activation_dropout=0.1)
    plan = Plan(st, 4, 8000, train=True, reg=reg)
    tr = SpeakerTrainer(st, plan, OneCycle(2e-3, 10))
    loss, _ = tr.train_step(wav, label, skip_layers=(1,))
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))
    assert float(st.mg("encoder.layers.1.feed_forward.output_dense.weight").abs().max()) == 0.0
    assert float(st.mg("encoder.layers.1.attention.out_proj.weight").abs().max()) == 0.0
    assert float(st.mg("encoder.layers.0.feed_forward.output_dense.weight").abs().max()) > 0.0
    assert float(st.mg("encoder.layers.2.layer_norm.weight").abs().max()) > 0.0       # produced by the skipped block's re-normalisation
    assert float(st.mg("masked_spec_embed").abs().max()) > 0.0
    losses = [float(loss)]
    for _ in range(4):
        l2, _ = tr.train_step(wav, label, skip_layers=())
        losses.append(float(l2))
    assert all(np.isfinite(losses)) and torch.isfinite(st.flat).all()
    if st.scaler is not None:
        assert int(st.scaler[3]) == 0                      # no step was skipped for an overflow
    if dtype != torch.float32:
        return
    # directional derivative with all dropout sites on (fixed step -> fixed masks), f32
    st2, _ = _store(cfg, ocfg, torch.float32, "aam", 10)
    plan2 = Plan(st2, 4, 8000, train=True, reg=dataclasses.replace(reg, layerdrop=0.0, mask_time_prob=0.0))

    def loss_at():
        plan2.embed(wav, None, (), 5)
        l, _ = plan2.head_forward_backward(label)
        return float(l.double())
    st2.zero_grad()
    plan2.embed(wav, None, (), 5)
    plan2.head_forward_backward(label)
    plan2.backward()
    torch.cuda.synchronize()
    gen = torch.Generator().manual_seed(9)
    names = [n for n in st2.shapes if st2.is_trainable(n) and n.startswith("wav2vec.model.encoder.layers.")]
    for name in (names[3], names[17], "wav2vec.model.encoder.layer_norm.weight", "wav2vec.model.encoder.layers.0.layer_norm.weight"):
        d = torch.randn(st2.shapes[name], generator=gen).to(DEV)
        d /= d.norm()
        ana = float((st2.g(name).double() * d.double()).sum())
        base = st2.p(name).clone()
        eps = 2e-2
        st2.p(name).copy_(base + eps * d)
        st2.sync_lowp()
        lp = loss_at()
        st2.p(name).copy_(base - eps * d)
        st2.sync_lowp()
        lm = loss_at()
        st2.p(name).copy_(base)
        st2.sync_lowp()
        num = (lp - lm) / (2 * eps)
        assert abs(num - ana) <= 3e-2 * max(abs(ana), abs(num)) + 2e-4, (name, num, ana)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unfrozen_cnn_every_gradient_vs_reference_golden(dtype):
    """completely_freeze_feature_extractor=False: gradients of ALL parameters, incl. the 7 conv layers and the
    layer-0 GroupNorm, against the reference golden (which kept the CNN trainable)."""
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    g = load("g1_tiny.npz")
    cfg, ocfg = _cfgs("tiny")
    st = ParamStore(cfg, DEV, dtype, head="aam", num_speakers=10, freeze_cnn=False)
    sd = O.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (10, 2 * cfg.hidden_size), 20211)
    st.load_state_dict(sd)
    assert st.n_train == st.n_total and [b[0] for b in st.grad_buckets()][-1] == "cnn"
    wav, label, mask = T(g["wav"]).to(DEV), T(g["label"]).to(DEV), T(g["mask"]).to(DEV)
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=_no_reg())
    st.zero_grad()
    plan.embed(wav, mask)
    loss, _ = plan.head_forward_backward(label)
    plan.backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert abs(float(loss) - float(g["loss"])) < (1e-4 if f32 else 5e-2) * abs(float(g["loss"]))
    gtol, floor = (2e-3, 1e-6) if f32 else (0.15, 3e-3)
    for name in st.shapes:
        key = "grad." + (name[len("wav2vec.model."):] if name.startswith("wav2vec.model.") else name)
        ref = g[key]
        err = np.linalg.norm(st.g(name).cpu().numpy().astype(np.float64) - ref)
        assert err <= gtol * np.linalg.norm(ref) + floor, (name, err, np.linalg.norm(ref))
    # one optimiser step moves the CNN weights too
    before = st.mp("feature_extractor.conv_layers.3.conv.weight").clone()
    st.adam_step(1e-3)
    assert not torch.equal(before, st.mp("feature_extractor.conv_layers.3.conv.weight"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_gradient_buckets_are_final_when_notified_under_layerdrop(dtype):
    """The overlapped all-reduce reads a bucket the moment Plan.backward notifies it: snapshot every bucket (raw and
    as merged by BucketAllReducer) at its notification, under LayerDrop patterns that exercise the paired /
    single / held weight-gradient launches and the deferred LayerNorm folds, and compare with the gradient after the
    whole backward -- bit for bit (the CPU twin of this test checks the schedule: test_host_cpu.py)."""
    import dataclasses
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.trainer import BucketAllReducer
    cfg, ocfg = _cfgs("tiny")
    cfg = dataclasses.replace(cfg, num_hidden_layers=6)
    ocfg = dataclasses.replace(ocfg, num_hidden_layers=6)
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0
    wav, label = O.synth_batch(4, 4000, 10, seed=3)
    wav, label = wav.to(DEV), label.to(DEV)
    plan = Plan(st, 4, 4000, train=True, reg=_no_reg())
    raw = {n: (s, e) for n, s, e in st.grad_buckets()}
    merged = BucketAllReducer(st, bucket_merge=2).ranges
    for skip in [(), (5,), (4,), (3,), (2, 3), (0,), (1, 4), (0, 1, 2, 3, 4, 5), (5, 3, 1), (4, 2, 0)]:
        st.zero_grad()
        plan.embed(wav, None, skip)
        plan.head_forward_backward(label)
        snaps, order = [], []

        def rec(name):
            order.append(name)
            for rng in (raw, merged):
                if name in rng:
                    s, e = rng[name]
                    snaps.append((name, s, e, st.grad[s:e].clone()))
        plan.backward(on_bucket_ready=rec)
        torch.cuda.synchronize()
        assert order == ["head"] + [f"layer{l}" for l in range(5, -1, -1)] + ["prologue", "projection"], order
        for name, s, e, snap in snaps:
            assert torch.equal(snap, st.grad[s:e]), (skip, name)
        for l in skip:                                   # LayerDrop: zero gradient
            s, e = raw[f"layer{l}"]
            assert float(st.grad[s:e].abs().max()) == 0.0
        assert torch.isfinite(st.grad).all()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_weight_gradient_groups_of_four_bit_equal_to_pairs_and_final_when_notified(dtype):
    """Round 6: the grouped weight-gradient launch may cover FOUR consecutive blocks (chosen for wav2vec2-large, where four
    fill the chip's rounds of 256 x 256 tiles and two do not; forced here with W2V2_WGRAD_GROUP on an 8-block tiny model).
    Same gradients bit for bit as with pairs -- the kernel writes each tile from its own K loop whatever the grouping --
    under LayerDrop patterns that close groups early, and every bucket is final when it is notified."""
    import dataclasses
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("tiny")
    cfg = dataclasses.replace(cfg, num_hidden_layers=8)
    ocfg = dataclasses.replace(ocfg, num_hidden_layers=8)
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0
    wav, label = O.synth_batch(4, 4000, 10, seed=3)
    wav, label = wav.to(DEV), label.to(DEV)
    raw = {n: (s, e) for n, s, e in st.grad_buckets()}
    old = os.environ.get("W2V2_WGRAD_GROUP")
    plans = {}
    try:
        for k in ("2", "4"):
            os.environ["W2V2_WGRAD_GROUP"] = k
            plans[k] = Plan(st, 4, 4000, train=True, reg=_no_reg())
            assert plans[k].wg_group == int(k) and len(plans[k]._gsets) == (8 if int(k) > 2 else 2)
    finally:
        if old is None:
            os.environ.pop("W2V2_WGRAD_GROUP", None)
        else:
            os.environ["W2V2_WGRAD_GROUP"] = old
    for skip in [(), (7,), (5,), (4, 3), (0,), (6, 2), (7, 6, 5, 4), (1, 3, 5, 7), (0, 1, 2, 3, 4, 5, 6, 7)]:
        got = {}
        for k, plan in plans.items():
            st.zero_grad()
            plan.embed(wav, None, skip)
            plan.head_forward_backward(label)
            snaps, order = [], []

            def rec(name):
                order.append(name)
                if name in raw:
                    s_, e_ = raw[name]
                    snaps.append((name, s_, e_, st.grad[s_:e_].clone()))
            plan.backward(on_bucket_ready=rec)
            torch.cuda.synchronize()
            assert order == ["head"] + [f"layer{l}" for l in range(7, -1, -1)] + ["prologue", "projection"], order
            for name, s_, e_, snap in snaps:
                assert torch.equal(snap, st.grad[s_:e_]), (k, skip, name)
            got[k] = st.grad.clone()
        assert torch.isfinite(got["4"][:st.n_train]).all()
        bad = [n for n in st.shapes if st.is_trainable(n) and not torch.equal(
            got["2"][st.offsets[n]:st.offsets[n] + st.g(n).numel()], got["4"][st.offsets[n]:st.offsets[n] + st.g(n).numel()])]
        assert not bad, (skip, bad[:5])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_partial_zero_grad_ranges_are_all_written_by_the_backward(dtype):
    """ParamStore.zero_grad(skip_layers) leaves the encoder's Linear gradients, the AAM weight and the pos-conv
    weight-norm pair UN-zeroed because the backward writes (does not accumulate into) them.  Enforce that coupling: poison
    the whole arena with NaN, run the partial zero, and every backward pattern used with it must leave a finite arena
    that is bitwise the gradient of a run that zeroed everything first."""
    import dataclasses
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("tiny")
    cfg = dataclasses.replace(cfg, num_hidden_layers=4)
    ocfg = dataclasses.replace(ocfg, num_hidden_layers=4)
    st, _ = _store(cfg, ocfg, dtype, "aam", 10)
    if st.scaler is not None:
        st.scaler[0] = 256.0
    wav, label = O.synth_batch(4, 4000, 10, seed=3)
    wav, label = wav.to(DEV), label.to(DEV)
    plan = Plan(st, 4, 4000, train=True, reg=_no_reg())
    assert plan.grouped                                   # the path whose zero_grad is partial
    for skip in [(), (3,), (0,), (1, 2), (0, 1, 2, 3)]:
        st.grad.zero_()
        plan.embed(wav, None, skip)
        plan.head_forward_backward(label)
        plan.backward()
        want = st.grad.clone()
        st.grad.fill_(float("nan"))
        st.zero_grad(tuple(skip))
        plan.embed(wav, None, skip)
        plan.head_forward_backward(label)
        plan.backward()
        torch.cuda.synchronize()
        bad = [n for n in st.shapes if st.is_trainable(n) and not torch.isfinite(st.g(n)).all()]
        assert not bad, (skip, bad)
        diff = [n for n in st.shapes if st.is_trainable(n) and not torch.equal(st.g(n), want[st.offsets[n]:st.offsets[n] + st.g(n).numel()].view_as(st.g(n)))]
        assert not diff, (skip, diff)


def test_large_24_layers_5s_batch32_properties():
    """BASELINE configs[3] at FULL depth and size (wav2vec2-large: 24 layers, H = 1024, 16 heads, FFN 4096; 5 s clips,
    T = 249; B = 32 per GPU; fp16 mode with dropout / LayerDrop-free regularisation for reproducibility).  The CPU
    oracle cannot run this in test time, so the checks are the size-independent properties: finite loss near
    ln(C) + margin effect, every gradient finite, two identical passes give bitwise-identical encoder gradients,
    LayerDrop-skipped layers get exactly zero gradient, and an utterance's eval embedding does not depend on
    its batch neighbours (bitwise: B = 32 vs B = 1)."""
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-large")
    assert cfg.num_hidden_layers == 24 and cfg.num_frames(80000) == 249
    B, N, C = 32, 80000, 5994
    st = ParamStore(cfg, DEV, torch.float16, head="aam", num_speakers=C, embed_dim=2 * cfg.hidden_size)
    st.init_weights(seed=3)
    wav, label = O.synth_batch(B, N, C, seed=5)
    wav, label = wav.to(DEV), label.to(DEV)
    reg = Wav2Vec2RegularisationConfig(mask_time_prob=0.0)          # dropout 0.1 stays on (counter-based: reproducible)
    tr = Plan(st, B, N, train=True, reg=reg, seed=11)
    skip = (3, 20)
    grads = []
    for _ in range(2):
        st.zero_grad()
        tr.embed(wav, None, skip, step=4)
        loss, sm = tr.head_forward_backward(label)
        tr.backward()
        torch.cuda.synchronize()
        grads.append(st.grad.clone())
    assert torch.isfinite(loss) and 5.0 < float(loss) < 30.0, float(loss)
    assert torch.isfinite(grads[0]).all()
    raw = {n: (s, e) for n, s, e in st.grad_buckets()}
    # the 24 encoder-layer buckets (grouped weight gradients, fixed-order LayerNorm folds) are bitwise reproducible;
    # the head (AAM column dots) and the prologue (pos-conv bias column sums) accumulate with f32 atomics
    for n, (s, e) in raw.items():
        a, b = grads[0][s:e], grads[1][s:e]
        if n.startswith("layer"):
            assert torch.equal(a, b), n
        else:
            assert float((a - b).norm()) <= 1e-5 * float(a.norm()), n
    for l in range(24):
        s, e = raw[f"layer{l}"]
        z = float(grads[0][s:e].abs().max())
        assert (z == 0.0) if l in skip else (z > 0.0), l
    del tr, grads
    ev = Plan(st, B, N, train=False)
    e32 = ev.embed(wav).clone()
    del ev
    ev1 = Plan(st, 1, N, train=False)
    for i in (0, 17, 31):
        e1 = ev1.embed(wav[i:i + 1])
        torch.cuda.synchronize()
        assert torch.equal(e1[0], e32[i]), i


def test_long_utterance_and_paired_model_run_on_the_tiled_attention():
    """every real test utterance (B = 1, up to ~145 s = 7249 frames, ref:
    speaker_recognition_module.py:462-500) and the paired model at its real size (T = 2*149+3 = 301, ref:
    wav2vec2_paired_input.py:163-207) take the fused (tiled) attention.  (i) a small d = 64 model on a 20 s utterance
    against the CPU oracle; (ii) w2v2-base on a 145 s utterance: fused vs the unfused path that materialises the
    [12, 7249, 7249] scores; (iii) the paired base model: logits, loss and every gradient bucket fused vs unfused."""
    import dataclasses
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    # (i) oracle parity at T = 999 (> 160: tiled kernels), fp16
    cfg = dataclasses.replace(W2V2Config.tiny(), hidden_size=128, num_attention_heads=2, intermediate_size=256)
    ocfg = dataclasses.replace(O.OracleConfig.tiny(), hidden_size=128, num_attention_heads=2, intermediate_size=256)
    N = 320000
    assert cfg.num_frames(N) == 999 and cfg.head_dim == 64
    st, sd = _store(cfg, ocfg, torch.float16, None, 1)
    wav, _ = O.synth_batch(1, N, 2, seed=8)
    ev = Plan(st, 1, N, train=False)
    assert ev.fused
    e = ev.embed(wav.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(e.cpu(), O.speaker_embedding(wav, sd, ocfg)) < 3e-3
    del ev
    # (ii) base model, 145 s
    cfg, ocfg = _cfgs("base")
    st, _ = _store(cfg, ocfg, torch.float16, None, 1)
    N = 2319840
    assert cfg.num_frames(N) == 7249
    wav, _ = O.synth_batch(1, N, 2, seed=9)
    ev = Plan(st, 1, N, train=False)
    assert ev.fused and ev.T == 7249
    e = ev.embed(wav.to(DEV)).clone()
    del ev
    un = Plan(st, 1, N, train=False, fused_attention=False)
    e2 = un.embed(wav.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(e).all() and rel_l2(e.cpu(), e2.cpu()) < 2e-3
    del un
    # (iii) paired-input model at its real size
    stp = ParamStore(cfg, DEV, torch.float16, head="bce")
    stp.init_weights(seed=4)
    stp.scaler[0] = 256.0
    B = 3
    wl, _ = O.synth_batch(2 * B, 48000, 2, seed=10)
    label = torch.tensor([1, 0, 1], device=DEV)
    res = []
    for fused in (True, False):
        plan = Plan(stp, B, 48000, train=True, reg=_no_reg(), pooling="first", paired=True, fused_attention=fused)
        assert plan.T == 301 and plan.fused == fused
        stp.zero_grad()
        plan.embed(wl[:, 0].to(DEV))
        loss, pred = plan.head_forward_backward(label)
        plan.backward()
        torch.cuda.synchronize()
        res.append((float(loss), pred.clone(), stp.grad.clone()))
        del plan
    assert abs(res[0][0] - res[1][0]) < 2e-3 * abs(res[1][0]) and torch.allclose(res[0][1], res[1][1], atol=2e-3)
    assert torch.isfinite(res[0][2]).all()
    for n, s_, e_ in stp.grad_buckets():
        a, b = res[0][2][s_:e_], res[1][2][s_:e_]
        assert float((a - b).norm()) < 3e-2 * float(b.norm()) + 1e-6, n


# ------------------------------------------------------------------------------------------------------------------
# Options of in-scope rows that round 2 guarded off (VERDICT r2 missing 4): activation_dropout > 0 with its backward,
# feature-axis SpecAugment (mask_feature_prob), IndexPool1D("random"), NoPooling.
def test_activation_dropout_and_feature_mask_backward_by_directional_derivatives():
    """ref: config/network/wav2vec2_fc.yaml:45,56 (activation_dropout, mask_feature_prob); HF:566-569, :1294-1304.
    Dropout keep decisions are a pure function of (seed, step, site, element), so with everything else fixed the loss is
    a deterministic differentiable function of the weights: in the exact-f32 mode the hand-written gradient must match
    central differences along random directions.  Also: masked feature channels are exactly zero after the projection,
    and the activation dropout really drops (a fraction ~p of FFN activations is zero)."""
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("tiny")
    B, N, C = 3, 4000, 10
    st, _ = _store(cfg, ocfg, torch.float32, "aam", C)
    reg = Wav2Vec2RegularisationConfig(activation_dropout=0.3, attention_dropout=0.0, feat_proj_dropout=0.0,
                                       hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.2)
    plan = Plan(st, B, N, train=True, reg=reg)
    wav, label = O.synth_batch(B, N, C, seed=4)
    wav, label = wav.to(DEV), label.to(DEV)
    g = torch.Generator().manual_seed(1)
    tmask = (torch.rand(B, plan.T0, generator=g) < 0.1).to(torch.uint8).to(DEV)
    fmask = (torch.rand(B, cfg.hidden_size, generator=g) < 0.2).to(torch.uint8).to(DEV)

    def loss_at():
        plan.embed(wav, tmask, (), 5, fmask)
        loss, _ = plan.head_forward_backward(label)
        return float(loss)

    st.zero_grad()
    l0 = loss_at()
    h0 = plan.h0.view(B, plan.T0, -1).float()
    assert float((h0 * fmask[:, None, :].float()).abs().max()) == 0.0 and float(h0.abs().max()) > 0
    hz = float((plan.lb[0].h == 0).float().mean())
    assert 0.2 < hz < 0.4, hz
    plan.backward()
    torch.cuda.synchronize()
    grad = st.grad.clone()
    names = ["wav2vec.model.encoder.layers.0.feed_forward.intermediate_dense.weight",
             "wav2vec.model.encoder.layers.1.feed_forward.output_dense.weight",
             "wav2vec.model.feature_projection.projection.weight", "wav2vec.model.masked_spec_embed",
             "wav2vec.model.encoder.layers.0.attention.q_proj.weight"]
    for i, name in enumerate(names):
        w = st.p(name)
        v = torch.randn(w.shape, generator=torch.Generator().manual_seed(10 + i)).to(DEV)
        v = v / v.norm() * w.norm() * 2e-3
        w0 = w.clone()
        w.copy_(w0 + v); st.sync_lowp(); lp = loss_at()
        w.copy_(w0 - v); st.sync_lowp(); lm = loss_at()
        w.copy_(w0); st.sync_lowp()
        fd = (lp - lm) / 2
        an = float((grad[st.offsets[name]:st.offsets[name] + w.numel()].view(w.shape) * v).sum())
        assert abs(fd - an) <= 0.03 * max(abs(an), abs(fd)) + 2e-6, (name, fd, an)
    assert abs(loss_at() - l0) < 1e-6            # same masks every time (counter-based dropout)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_random_index_pooling_and_no_pooling(dtype):
    """ref: src/layers/pooling.py:125-126,150-166 + speaker_recognition_module.py:246-267.  "random": the pooled
    embedding is frame t* of the encoder output, t* drawn per forward call; its gradient lands on that frame only.
    "none": every frame is an embedding; CE over B * T rows with the utterance label repeated per frame, checked against
    torch on the engine's own embeddings; d(loss)/d(hidden) equals autograd's."""
    from w2v2_speaker_amd.engine import Plan
    cfg, ocfg = _cfgs("tiny")
    B, N, C = 3, 4000, 10
    f32 = dtype == torch.float32
    wav, label = O.synth_batch(B, N, C, seed=9)
    wav, label = wav.to(DEV), label.to(DEV)
    st, _ = _store(cfg, ocfg, dtype, "aam", C, embed_dim=cfg.hidden_size)
    if st.scaler is not None:
        st.scaler[0] = 64.0
    plan = Plan(st, B, N, train=True, reg=_no_reg(), pooling="random")
    seen = set()
    for _ in range(6):
        emb = plan.embed(wav).clone()
        t = plan._rand_idx
        seen.add(t)
        assert torch.equal(emb, plan.out[:, t, :].float())
    assert len(seen) > 1 and all(0 <= t < plan.T for t in seen)
    st.zero_grad()
    plan.head_forward_backward(label)
    demb = plan.demb.clone()
    from w2v2_speaker_amd import ops
    ops.pool_bwd(plan.out, plan.emb, demb, plan.G.view(B, plan.T, -1), ops.POOL_INDEX_BASE + plan._rand_idx)
    G = plan.G.view(B, plan.T, -1).float()
    assert torch.allclose(G[:, plan._rand_idx], demb, rtol=2e-3 if not f32 else 0, atol=0 if f32 else 1e-2 * float(demb.abs().max()))
    G[:, plan._rand_idx] = 0
    assert float(G.abs().max()) == 0.0
    # ---- no pooling, CE head on every frame
    st2, _ = _store(cfg, ocfg, dtype, "ce", C, embed_dim=cfg.hidden_size)
    if st2.scaler is not None:
        st2.scaler[0] = 64.0
    p2 = Plan(st2, B, N, train=True, reg=_no_reg(), pooling="none")
    emb = p2.embed(wav)
    T = p2.T
    assert emb.shape == (B * T, cfg.hidden_size) and torch.equal(emb.view(B, T, -1), p2.out.float())
    st2.zero_grad()
    loss, sm = p2.head_forward_backward(label)
    gs = _gscale(st2)
    e = emb.detach().clone().requires_grad_(True)
    W, b = st2.p("fc_list.0.0.weight"), st2.p("fc_list.0.0.bias")
    if not f32:
        logits = e.to(dtype).float() @ W.to(dtype).float().t() + b
    else:
        logits = e @ W.t() + b
    ref = torch.nn.functional.cross_entropy(logits, label.repeat_interleave(T))
    ref.backward()
    assert sm.shape == (B * T, C)
    assert abs(float(loss) - float(ref)) < (1e-5 if f32 else 3e-3) * abs(float(ref))
    assert rel_l2((p2.demb / gs).cpu(), e.grad.cpu()) < (1e-4 if f32 else 1e-2)
    p2.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(st2.grad).all() and float(st2.g("wav2vec.model.encoder.layers.0.attention.q_proj.weight").abs().max()) > 0


@pytest.mark.parametrize("where", ["head", "layer1", "layer0", "prologue"])
def test_overflow_anywhere_in_the_backward_chain_reaches_the_scanned_bucket(where):
    """found_inf is decided by scanning only the LAST gradient bucket backward writes (the prologue), which
    relies on a non-finite activation gradient surviving every kernel below the point where it appears -- residual
    joins, LayerNorm backward, attention backward, SpecAugment mask backward, dropout, LayerDrop-skipped layers.  An inf
    is injected into the running activation gradient at the top of the chain, in front of each layer and in front of
    the prologue, with dropout / time masks on and one layer skipped: the step must be skipped and the scale halved
    every time, and the parameters must stay finite."""
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import Constant
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    import dataclasses
    cfg, ocfg = _cfgs("tiny")
    cfg = dataclasses.replace(cfg, num_hidden_layers=3)
    from w2v2_speaker_amd.params import ParamStore
    st = ParamStore(cfg, DEV, torch.float16, head="aam", num_speakers=10)
    st.init_weights(3)
    st.scaler[0] = 256.0
    reg = Wav2Vec2RegularisationConfig(attention_dropout=0.1, feat_proj_dropout=0.1, hidden_dropout=0.1, layerdrop=0.0,
                                       mask_time_prob=0.3, mask_time_length=2)
    plan = Plan(st, 4, 4000, train=True, reg=reg)
    tr = SpeakerTrainer(st, plan, Constant(1e-3))
    wav, label = O.synth_batch(4, 4000, 10, seed=2)
    wav, label = wav.to(DEV), label.to(DEV)
    tr.train_step(wav, label, skip_layers=())                       # a clean step first
    torch.cuda.synchronize()
    assert int(st.scaler[3]) == 0
    p_before = st.flat.clone()
    body, head_fb = plan._layer_backward_body, plan.head_forward_backward
    if where == "head":
        def poisoned(lbl):
            out = head_fb(lbl)
            plan.demb[1, 3] = float("inf")
            return out
        plan.head_forward_backward = poisoned
    elif where.startswith("layer"):
        target = int(where[5:])

        def poked(l, lnfold):
            if l == target:
                plan.G.view(-1)[4321 % plan.G.numel()] = float("-inf")
            return body(l, lnfold)
        plan._layer_backward_body = poked
    else:
        def last(l, lnfold):
            r = body(l, lnfold)
            if l == 0:
                plan.G.view(-1)[777] = float("nan")           # after the last layer body: only the prologue is below
            return r
        plan._layer_backward_body = last
    tr.train_step(wav, label, skip_layers=(2,) if where != "layer1" else (0,))
    torch.cuda.synchronize()
    assert int(st.scaler[3]) == 1 and float(st.scaler[0]) == 128.0, (where, st.scaler.tolist())
    assert torch.equal(st.flat, p_before), "a skipped step must not touch the parameters"
    plan._layer_backward_body, plan.head_forward_backward = body, head_fb
    tr.train_step(wav, label, skip_layers=())
    torch.cuda.synchronize()
    assert int(st.scaler[3]) == 1 and torch.isfinite(st.flat).all() and not torch.equal(st.flat, p_before)


SB_DIR = os.environ.get("W2V2_SB_GOLDEN_DIR", GOLDEN)


@pytest.mark.skipif(not os.path.exists(os.path.join(SB_DIR, "g15_sb_asp.npz")),
                    reason="no speechbrain golden (tests/golden/make_sb_goldens.py needs speechbrain): row a10 stays unpinned")
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_attentive_pooling_vs_speechbrain_golden(dtype):
    """SURVEY 8a row a10 against the REAL speechbrain AttentiveStatisticsPooling(768) (g15_sb_asp.npz: C = 768, T = 149,
    B = 4, training mode; ref: src/layers/pooling.py:87-106): output, input gradient and the six parameter gradients of the
    HIP kernels (csrc/asp.hip + asp.py) for the golden's upstream gradient."""
    from w2v2_speaker_amd.asp import ASP_PREFIX, AttentivePool
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore
    g = np.load(os.path.join(SB_DIR, "g15_sb_asp.npz"), allow_pickle=False)
    B, Tn, C = g["x"].shape
    st = ParamStore(W2V2Config(), DEV, dtype, head="aam", num_speakers=10, attentive_pool=True)
    st.init_weights(seed=1)
    st.load_state_dict({ASP_PREFIX + k[len("param."):]: T(g[k]) for k in g.files if k.startswith("param.")}, strict=False)
    rows = (B * Tn + 63) // 64 * 64
    full = torch.zeros(rows, C, dtype=st.act_dtype, device=DEV)
    x = full[:B * Tn]
    x._w2v2_padded = full
    x.copy_(T(g["x"]).view(B * Tn, C).to(DEV))
    emb = torch.empty(B, 2 * C, dtype=torch.float32, device=DEV)
    dx = torch.zeros(B * Tn, C, dtype=st.act_dtype, device=DEV)
    pool = AttentivePool(st, x, emb, dx, B, Tn, True)
    st.zero_grad()
    pool.forward()
    pool.backward(T(g["upstream"]).to(DEV))
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert rel_l2(emb.cpu(), g["out"]) < (2e-5 if f32 else 3e-3)
    assert rel_l2(dx.float().cpu().view(B, Tn, C), g["dx"]) < (1e-3 if f32 else 3e-2)
    for k in g.files:
        if k.startswith("grad."):
            got = st.g(ASP_PREFIX + k[len("grad."):]).cpu().reshape(g[k].shape)
            assert rel_l2(got, g[k]) < (2e-3 if f32 else 5e-2), k
