#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (authoring container only).

Imports /root/reference (read-only) with the three shims of SURVEY.md App. C, builds the HF
``Wav2Vec2Model`` through the reference's own ``Wav2Vec2WrapperModule``, overwrites every parameter
from the name-keyed PCG64 generator in ``oracle.w2v2_oracle.synth_tensor`` and dumps small input /
expected-output vectors.  The fixtures are data only; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)

# ----------------------------------------------------------------------------- shims
REF = "/root/reference"
sys.path.insert(0, REF)
torch.cuda.Device = torch.device                                   # src/models/wav2vec2.py:28
pl = types.ModuleType("pytorch_lightning")


class LightningModule(torch.nn.Module):
    pass


pl.LightningModule, pl.Trainer = LightningModule, object
sys.modules["pytorch_lightning"] = pl
for n in ("hurry", "hurry.filesize"):
    sys.modules[n] = types.ModuleType(n)
sys.modules["hurry.filesize"].size = lambda *a, **k: ""
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))     # debug plots only
sb = "speechbrain.lobes.models.ECAPA_TDNN"
parts = sb.split(".")
for i in range(1, 5):
    sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
sys.modules[sb].AttentiveStatisticsPooling = object

from transformers import Wav2Vec2Config, Wav2Vec2Model  # noqa: E402
from transformers.models.wav2vec2 import modeling_wav2vec2 as hf_mod  # noqa: E402

from oracle import w2v2_oracle as O  # noqa: E402

_CFG_EXTRA = {}


def _from_config(hid, **ov):
    ov.pop("gradient_checkpointing", None)
    return Wav2Vec2Model(Wav2Vec2Config(attn_implementation="eager", **_CFG_EXTRA, **ov))


Wav2Vec2Model.from_pretrained = staticmethod(_from_config)

from src.models.wav2vec2 import Wav2Vec2WrapperModule, Wav2Vec2RegularisationConfig  # noqa: E402
from src.layers.pooling import (MeanStdStatPool1D, MeanStatPool1D, IndexPool1D, MaxPool1D,  # noqa: E402
                                QuantilePool1D)
from src.optim.loss import AngularAdditiveMarginSoftMaxLoss, CrossEntropyLoss  # noqa: E402
from src.eval_metrics import calculate_eer, calculate_mdc  # noqa: E402
from src.evaluation.speaker.cosine_distance import compute_cosine_scores  # noqa: E402
from src.data.preprocess.input_normalisation import InputNormalizer2D  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def hf_extra(cfg: O.OracleConfig):
    return dict(conv_dim=list(cfg.conv_dim), hidden_size=cfg.hidden_size,
                num_hidden_layers=cfg.num_hidden_layers,
                num_attention_heads=cfg.num_attention_heads,
                intermediate_size=cfg.intermediate_size,
                num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups,
                do_stable_layer_norm=cfg.do_stable_layer_norm, feat_extract_norm=cfg.feat_extract_norm,
                conv_bias=cfg.conv_bias)


def build_reference_wrapper(cfg: O.OracleConfig, seed: int, cls_token=False, hf_id="facebook/wav2vec2-base", family=None):
    """The reference's wrapper (src/models/wav2vec2.py:97-146) with dropouts/layerdrop/masking = 0.  ``hf_id`` is what the
    wrapper inspects for "base" / "large" (:112-117); ``family="outlier"`` = data/synthetic.py outlier_family on the weights."""
    global _CFG_EXTRA
    _CFG_EXTRA = hf_extra(cfg)
    reg = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0,
                                       feat_proj_dropout=0.0, hidden_dropout=0.0, layerdrop=0.0,
                                       mask_time_prob=0.05)   # keep masked_spec_embed alive
    w = Wav2Vec2WrapperModule(hf_id, False, reg, insert_clc_token=cls_token)
    sd = O.make_state_dict(cfg, seed)
    if family == "outlier":
        from w2v2_speaker_amd.data.synthetic import outlier_family
        sd = {k: torch.from_numpy(v) for k, v in outlier_family({k: v.numpy() for k, v in sd.items()}, seed).items()}
    missing, unexpected = w.model.load_state_dict(sd, strict=True), None
    return w, sd


def to_np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in d.items()}


# ----------------------------------------------------------------------------- G1 tiny, all stages + grads
def golden_tiny():
    cfg = O.OracleConfig.tiny()
    B, N, C = 2, 4000, 10
    w, sd = build_reference_wrapper(cfg, seed=20211)
    wav, label = O.synth_batch(B, N, C, seed=42133724)
    T = cfg.num_frames(N)
    rng = np.random.Generator(np.random.PCG64(7))
    mask = np.zeros((B, T), dtype=bool)
    for b in range(B):
        s = rng.integers(0, T - 3)
        mask[b, s:s + 3] = True
    mask_t = torch.from_numpy(mask)

    # the ~40 lines of tensor logic of Wav2vec2FCModule (wav2vec2_fc.py:414-438) restated, since the
    # LightningModule itself needs PL/torchmetrics: squeeze -> wrapper -> transpose -> pool
    w.train()                       # dropouts are all 0; mask injected
    stages = {}
    hooks = []

    def grab(name):
        def f(_m, _i, o):
            stages[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
        return f
    m = w.model
    hooks.append(m.feature_extractor.register_forward_hook(grab("conv_out_bct")))
    hooks.append(m.feature_projection.register_forward_hook(grab("proj")))
    hooks.append(m.encoder.pos_conv_embed.register_forward_hook(grab("pos_conv")))
    hooks.append(m.encoder.layer_norm.register_forward_hook(grab("enc_in")))
    for l, layer in enumerate(m.encoder.layers):
        hooks.append(layer.register_forward_hook(grab(f"layer{l}")))
    x = torch.squeeze(wav)                                    # wav2vec2_fc.py:418-419
    out = m(x, mask_time_indices=mask_t).last_hidden_state.transpose(1, 2)   # models/wav2vec2.py:71-74
    emb_in = out.transpose(2, 1)                              # wav2vec2_fc.py:428
    emb = MeanStdStatPool1D(dim_to_reduce=1)(emb_in)
    loss_fn = AngularAdditiveMarginSoftMaxLoss(2 * cfg.hidden_size, C, margin=0.2, scale=30)
    aam_w = O.synth_tensor("loss_fn.fc_weights", (C, 2 * cfg.hidden_size), 20211)
    with torch.no_grad():
        loss_fn.fc_weights.copy_(aam_w)
    loss, pred = loss_fn(emb, label)
    loss.backward()
    for h in hooks:
        h.remove()
    g = {"wav": wav, "label": label, "mask": mask_t, "embedding": emb, "loss": loss,
         "softmax": pred, "last_hidden": emb_in,
         "stage.conv_out": stages["conv_out_bct"].transpose(1, 2), "stage.proj": stages["proj"],
         "stage.pos_conv": stages["pos_conv"], "stage.enc_in": stages["enc_in"]}
    for l in range(cfg.num_hidden_layers):
        g[f"stage.layer{l}"] = stages[f"layer{l}"]
    for n, p in m.named_parameters():
        g["grad." + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    g["grad.loss_fn.fc_weights"] = loss_fn.fc_weights.grad

    # CE head (config 1): nn.Linear + CrossEntropyLoss (wav2vec2_fc.py:199-210, cross_entropy.py:27-31)
    ce_w = O.synth_tensor("fc_list.0.0.weight", (C, 2 * cfg.hidden_size), 20211)
    ce_b = O.synth_tensor("fc_list.0.0.bias", (C,), 20211)
    ce_loss, ce_pred = CrossEntropyLoss()(emb.detach() @ ce_w.t() + ce_b, label)
    g["ce.loss"], g["ce.softmax"] = ce_loss, ce_pred

    # eval-mode embeddings of the other pools (f1 rows)
    w.eval()
    with torch.no_grad():
        h = w(x).transpose(2, 1)
        g["eval.last_hidden"] = h
        g["eval.mean+std"] = MeanStdStatPool1D(1)(h)
        g["eval.mean"] = MeanStatPool1D(1)(h)
        g["eval.max"] = MaxPool1D(1)(h)
        g["eval.first"] = IndexPool1D("first", 1)(h)
        g["eval.middle"] = IndexPool1D("middle", 1)(h)
        g["eval.last"] = IndexPool1D("last", 1)(h)
        g["eval.quantile"] = QuantilePool1D(1)(h)
    wc, _ = build_reference_wrapper(cfg, seed=20211, cls_token=True)
    wc.eval()
    # the reference hard-codes 768 for the CLS token (quirk Q5): patch the size for the tiny model only
    import src.models.wav2vec2 as refw
    orig_ones = refw.t.ones
    refw.t.ones = lambda shape, **kw: orig_ones((shape[0], shape[1], cfg.hidden_size), **kw)
    try:
        with torch.no_grad():
            hc = wc(x).transpose(2, 1)
    finally:
        refw.t.ones = orig_ones
    g["eval.cls.last_hidden"] = hc
    g["eval.first+cls"] = IndexPool1D("first+cls", 1)(hc)
    np.savez_compressed(os.path.join(OUT, "g1_tiny.npz"), **to_np(g))
    print("g1_tiny: loss", float(loss), "T", T, "keys", len(g))


# ----------------------------------------------------------------------------- G2/G3 base config
def golden_base():
    cfg = O.OracleConfig.base()
    B, N = 2, 48000
    w, sd = build_reference_wrapper(cfg, seed=20211)
    wav, label = O.synth_batch(B, N, 5994, seed=42133724)
    x = torch.squeeze(wav)
    g = {"label": label}
    w.eval()
    with torch.no_grad():
        h = w(x).transpose(2, 1)
        g["eval.mean+std"] = MeanStdStatPool1D(1)(h)
        g["eval.last_hidden.sample"] = h[:, ::16, ::16].contiguous()
    wc, _ = build_reference_wrapper(cfg, seed=20211, cls_token=True)
    wc.eval()
    with torch.no_grad():
        hc = wc(x).transpose(2, 1)
        g["eval.first+cls"] = IndexPool1D("first+cls", 1)(hc)

    # G3: train mode, injected mask, dropouts/layerdrop 0, AAM C=5994 and CE C=1211
    T = cfg.num_frames(N)
    np.random.seed(7)
    mask = hf_mod._compute_mask_indices((B, T), mask_prob=0.05, mask_length=10, min_masks=2)
    mask_t = torch.from_numpy(mask)
    g["mask"] = mask_t
    w.train()
    m = w.model
    out = m(x, mask_time_indices=mask_t).last_hidden_state
    emb = MeanStdStatPool1D(1)(out)
    loss_fn = AngularAdditiveMarginSoftMaxLoss(1536, 5994, margin=0.2, scale=30)
    with torch.no_grad():
        loss_fn.fc_weights.copy_(O.synth_tensor("loss_fn.fc_weights", (5994, 1536), 20211))
    loss, pred = loss_fn(emb, label)
    loss.backward()
    g["train.embedding"] = emb
    g["train.loss"] = loss
    g["train.softmax.sample"] = pred[:, ::37].contiguous()
    g["train.softmax.label"] = pred.gather(1, label.view(-1, 1))
    names, norms = [], []
    for n, p in list(m.named_parameters()) + [("loss_fn.fc_weights", loss_fn.fc_weights)]:
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(n)
        norms.append(float(gr.double().norm()))
        g["gradhead." + n] = gr.flatten()[:32].clone()
    g["grad_names"] = np.array(names)
    g["grad_norms"] = np.array(norms)
    lab_ce = label % 1211
    ce_w = O.synth_tensor("fc_list.0.0.weight", (1211, 1536), 20211)
    ce_b = O.synth_tensor("fc_list.0.0.bias", (1211,), 20211)
    ce_loss, ce_pred = CrossEntropyLoss()(emb.detach() @ ce_w.t() + ce_b, lab_ce)
    g["ce.loss"], g["ce.softmax.label"] = ce_loss, ce_pred.gather(1, lab_ce.view(-1, 1))
    np.savez_compressed(os.path.join(OUT, "g2_base.npz"), **to_np(g))
    print("g2_base: loss", float(loss), "emb norm", float(emb.norm()))


# ----------------------------------------------------------------------------- G10 second base golden
def golden_base2():
    """A second, independent data point for the fp16 embedding bound (the two-term-weight design
    was tuned on g2_base): other weight seed, other utterances, batch 8, 5 s clips (T = 249: other attention tiling,
    other GEMM row counts).  Eval mode, mean+std pooling -- the quantity the 1e-3 rel-L2 target is stated on."""
    cfg = O.OracleConfig.base()
    B, N = 8, 80000
    w, _ = build_reference_wrapper(cfg, seed=777)
    wav, _ = O.synth_batch(B, N, 5994, seed=31337)
    x = torch.squeeze(wav)
    w.eval()
    with torch.no_grad():
        h = w(x).transpose(2, 1)
        g = {"eval.mean+std": MeanStdStatPool1D(1)(h), "eval.last_hidden.sample": h[:, ::16, ::16].contiguous()}
    np.savez_compressed(os.path.join(OUT, "g10_base2.npz"), **to_np(g))
    print("g10_base2: emb norm", float(g["eval.mean+std"].norm()), "T", h.shape[1])


# ----------------------------------------------------------------------------- G11 base config at the benchmark's batch size
def golden_base66():
    """BASELINE configs[1] at ITS OWN size: 66 utterances of 3 s through the reference's
    ``Wav2Vec2WrapperModule`` (eval mode) + mean+std pooling, the weights of g2_base (seed 20211), the batch bench.py
    steps on (``synth_batch(66, 48000, 5994, seed=42133724)``: rows 0 / 1 are g2_base's two utterances).  Stored:
    the [66, 1536] embeddings (f32, 400 KB) and a strided sample of the hidden states."""
    cfg = O.OracleConfig.base()
    B, N = 66, 48000
    w, _ = build_reference_wrapper(cfg, seed=20211)
    wav, _ = O.synth_batch(B, N, 5994, seed=42133724)
    x = torch.squeeze(wav)
    w.eval()
    embs, samples = [], []
    with torch.no_grad():
        for i in range(0, B, 6):                   # (chunks of 6: the reference's forward is batch-independent in eval)
            h = w(x[i:i + 6]).transpose(2, 1)
            embs.append(MeanStdStatPool1D(1)(h))
            samples.append(h[:, ::32, ::32].contiguous())
    g = {"eval.mean+std": torch.cat(embs), "eval.last_hidden.sample": torch.cat(samples)}
    np.savez_compressed(os.path.join(OUT, "g11_base66.npz"), **to_np(g))
    print("g11_base66: emb norm", float(g["eval.mean+std"].norm()), "rows", g["eval.mean+std"].shape)


# ----------------------------------------------------------------------------- G4 AAM known answers
def golden_aam():
    g = {}
    D, C, B = 24, 7, 9
    rng = np.random.Generator(np.random.PCG64(4))
    W = torch.from_numpy(rng.standard_normal((C, D)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    label = torch.from_numpy(rng.integers(0, C, size=(B,)).astype(np.int64))
    x[0] = -3.0 * W[label[0]]                 # cos = -1  -> (cos - th) <= 0 branch, clamp edge
    x[1] = 2.0 * W[label[1]]                  # cos = +1  -> sine clamp edge
    x[2] = -W[label[2]] + 0.05 * x[2]         # cos slightly above -1, still below th=-0.98
    for margin, scale in ((0.2, 30.0), (0.3, 15.0)):
        fn = AngularAdditiveMarginSoftMaxLoss(D, C, margin=margin, scale=scale)
        with torch.no_grad():
            fn.fc_weights.copy_(W)
        xi = x.clone().requires_grad_(True)
        loss, pred = fn(xi, label)
        loss.backward()
        k = f"m{margin}_s{scale}."
        g[k + "loss"], g[k + "softmax"] = loss, pred
        g[k + "dx"], g[k + "dW"] = xi.grad, fn.fc_weights.grad
    # easy_margin=True (ref: aam_softmax.py:60-61): phi where cos > 0, the cosine itself elsewhere.  Rows 0 / 2 sit at
    # cos ~ -1 (the `cosine` branch), row 1 at cos = +1, the random rows on both sides of 0.
    fn = AngularAdditiveMarginSoftMaxLoss(D, C, margin=0.2, scale=30.0, easy_margin=True)
    with torch.no_grad():
        fn.fc_weights.copy_(W)
    xi = x.clone().requires_grad_(True)
    loss, pred = fn(xi, label)
    loss.backward()
    g["easy_m0.2_s30.0.loss"], g["easy_m0.2_s30.0.softmax"] = loss, pred
    g["easy_m0.2_s30.0.dx"], g["easy_m0.2_s30.0.dW"] = xi.grad, fn.fc_weights.grad
    with torch.no_grad():
        xn = torch.nn.functional.normalize(x) @ torch.nn.functional.normalize(W).t()
        g["easy.label_cos"] = xn.gather(1, label.view(-1, 1))[:, 0]
    g.update(x=x, W=W, label=label)
    np.savez_compressed(os.path.join(OUT, "g4_aam.npz"), **to_np(g))
    print("g4_aam ok")


# ----------------------------------------------------------------------------- G5 pooling
def golden_pool():
    g = {}
    rng = np.random.Generator(np.random.PCG64(5))
    for name, shape in (("small", (3, 11, 20)), ("t1", (2, 1, 8)), ("long", (1, 7249, 16))):
        x = torch.from_numpy((rng.standard_normal(shape) * 2 + 0.5).astype(np.float32))
        xi = x.clone().requires_grad_(True)
        y = MeanStdStatPool1D(1)(xi)
        g[name + ".x"], g[name + ".mean+std"] = x, y
        if shape[1] > 1:
            up = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
            (y * up).sum().backward()
            g[name + ".upstream"], g[name + ".dx"] = up, xi.grad
        g[name + ".mean"] = MeanStatPool1D(1)(x)
        g[name + ".max"] = MaxPool1D(1)(x)
        g[name + ".first"] = IndexPool1D("first", 1)(x)
        g[name + ".middle"] = IndexPool1D("middle", 1)(x)
    np.savez_compressed(os.path.join(OUT, "g5_pool.npz"), **to_np(g))
    print("g5_pool ok")


# ----------------------------------------------------------------------------- G6/G7 EER, minDCF, cosine
def golden_eval():
    g = {}
    rng = np.random.Generator(np.random.PCG64(6))
    n = 1000
    gt = rng.integers(0, 2, size=n)
    sc = np.clip(0.5 + 0.18 * rng.standard_normal(n) + 0.15 * (gt - 0.5), 0, 1)
    eer, thr = calculate_eer(gt.tolist(), sc.tolist())
    mdc, mthr = calculate_mdc(gt.tolist(), sc.tolist())
    g.update(gt=gt, scores=sc, eer=eer, eer_thr=thr, mdc=mdc, mdc_thr=mthr)
    a = torch.from_numpy(rng.standard_normal((50, 32)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal((50, 32)).astype(np.float32))
    b[:5] = a[:5] * 2
    b[5:8] = -a[5:8]
    cos = np.array(compute_cosine_scores(a, b))
    g.update(cos_a=a, cos_b=b, cos=cos, cos01=np.clip((cos + 1) / 2, 0, 1))   # evaluator.py:81
    # input normaliser (input_normalisation.py:54-67)
    wav = torch.from_numpy((rng.standard_normal((1, 4000)) * 0.1 + 0.02).astype(np.float32))
    g["norm_in"] = wav
    g["norm_out"] = InputNormalizer2D.normalize(wav, False)[0]   # normalize_over_channels: false
    np.savez_compressed(os.path.join(OUT, "g6_eval.npz"), **to_np(g))
    print("g6_eval: eer", eer, "mdc", mdc)


# ----------------------------------------------------------------------------- G8 OneCycleLR + Adam, masks
def golden_optim():
    g = {}
    p = torch.nn.Parameter(torch.linspace(-1, 1, 16))
    opt = torch.optim.Adam([p], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=100, div_factor=25)
    rng = np.random.Generator(np.random.PCG64(8))
    lrs, b1s, ps, gs = [], [], [], []
    for i in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        b1s.append(opt.param_groups[0]["betas"][0])
        gr = torch.from_numpy(rng.standard_normal(16).astype(np.float32))
        p.grad = gr.clone()
        opt.step()
        sch.step()
        gs.append(gr.numpy())
        ps.append(p.detach().clone().numpy())
    g.update(lr=np.array(lrs), beta1=np.array(b1s), grads=np.array(gs), params=np.array(ps),
             p0=torch.linspace(-1, 1, 16).numpy())
    # SpecAugment time masks from HF's sampler (HF:101-217) under np.random.seed
    for seed, shape in ((7, (66, 149)), (11, (4, 249))):
        np.random.seed(seed)
        g[f"mask.seed{seed}"] = hf_mod._compute_mask_indices(shape, mask_prob=0.05, mask_length=10,
                                                            min_masks=2)
    np.savez_compressed(os.path.join(OUT, "g8_optim.npz"), **to_np(g))
    print("g8_optim ok")


def golden_bce():
    """ref: src/optim/loss/binary_cross_entropy.py:24-40 (``BinaryCrossEntropyLoss``, the loss of the paired-input
    model) run as is: loss, prediction and d(loss)/d(logits) on seeded logits incl. saturated ones (|logit| = 40:
    the numerically stable branch) and both label values; [B, 1] logits like the module's nn.Linear(H, 1) output."""
    from src.optim.loss.binary_cross_entropy import BinaryCrossEntropyLoss
    rng = np.random.Generator(np.random.PCG64(9))
    logits = rng.standard_normal((24, 1)).astype(np.float32) * 3.0
    logits[:4, 0] = [40.0, -40.0, 0.0, 17.5]
    label = rng.integers(0, 2, size=(24,)).astype(np.int64)
    label[:4] = [0, 1, 1, 1]
    lg = torch.from_numpy(logits).requires_grad_(True)
    loss, pred = BinaryCrossEntropyLoss()(lg, torch.from_numpy(label))
    loss.backward()
    g = {"logits": logits, "label": label, "loss": loss.detach(), "prediction": pred, "dlogits": lg.grad}
    np.savez_compressed(os.path.join(OUT, "g9_bce.npz"), **to_np(g))
    print("g9_bce: loss", float(loss))


# ----------------------------------------------------------------------------- G12 the "eval EER" half of the metric
def golden_eer(mix=None):
    """a structured synthetic trial set (w2v2_speaker_amd/data/synthetic.py: 8 speakers x 4
    utterances of 3 s, speaker "voice" + fresh noise) through the reference's wrapper (eval mode, the g2_base weights)
    + mean+std pooling, then through the reference's OWN evaluator (CosineDistanceEvaluator(False, False, 0).evaluate:
    ref src/evaluation/speaker/speaker_recognition_evaluator.py:46-115, cosine_distance.py:107-132) on all 496 pairs.
    Stored: the [32, 1536] reference embeddings, the [0,1] scores the reference feeds calculate_eer, its EER / minDCF."""
    from src.evaluation.speaker.cosine_distance import CosineDistanceEvaluator
    from src.evaluation.speaker.speaker_recognition_evaluator import EvaluationPair, EmbeddingSample
    from w2v2_speaker_amd.data.synthetic import TRIAL_SET_DEFAULT, synth_trial_set
    kw = dict(TRIAL_SET_DEFAULT)
    if mix is not None:
        kw["mix"] = mix
    wav, spk, keys, trials = synth_trial_set(**kw)
    cfg = O.OracleConfig.base()
    w, _ = build_reference_wrapper(cfg, seed=20211)
    w.eval()
    x = torch.from_numpy(wav)
    embs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], 8):
            embs.append(MeanStdStatPool1D(1)(w(x[i:i + 8]).transpose(2, 1)))
    emb = torch.cat(embs)
    pairs = [EvaluationPair(bool(same), keys[i], keys[j]) for same, i, j in trials]
    samples = [EmbeddingSample(k, e) for k, e in zip(keys, emb)]
    ev = CosineDistanceEvaluator(False, False, 0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        res = ev.evaluate(pairs, samples)
    scores = np.clip((np.array(ev._compute_prediction_scores(
        [(samples[i], samples[j]) for _, i, j in trials])) + 1) / 2, 0, 1)
    g = {"embedding": emb, "speaker": spk, "trials": np.array(trials, dtype=np.int64), "scores": scores,
         "eer": res["eer"], "eer_threshold": res["eer_threshold"], "mdc": res["mdc"], "mdc_threshold": res["mdc_threshold"],
         "params": np.array([kw["n_speakers"], kw["utts_per_speaker"], kw["n_samples"], kw["seed"]], dtype=np.int64),
         "mix": np.float64(kw["mix"])}
    # the evaluator's NON-default branches, again through the reference's own code.
    # (a) centring (ref speaker_recognition_evaluator.py:154-172 = per-dimension z-score with the statistics of
    # fit_parameters) with and without length norm, fitted on the 32 reference embeddings themselves;
    # (b) length norm alone; (c) the non-pooled scoring of 2-D embeddings (ref cosine_distance.py:203-232) on six
    # seeded [frames, 64] embeddings (15 pairs; frame counts either side of the 50-frame subsample) with
    # ``random.seed(1234)`` set right before the call.
    prs = [(samples[i], samples[j]) for _, i, j in trials]
    for tag, (cen, ln) in {"c": (True, False), "cl": (True, True), "l": (False, True)}.items():
        ev2 = CosineDistanceEvaluator(cen, ln, 32)
        ev2.fit_parameters([e for e in emb], [])
        with contextlib.redirect_stdout(io.StringIO()):
            r2 = ev2.evaluate(pairs, samples)
            sc2 = np.clip((np.array(ev2._compute_prediction_scores(prs)) + 1) / 2, 0, 1)
        g[f"scores_{tag}"] = sc2
        g[f"eer_{tag}"], g[f"mdc_{tag}"] = r2["eer"], r2["mdc"]
        g[f"eer_threshold_{tag}"], g[f"mdc_threshold_{tag}"] = r2["eer_threshold"], r2["mdc_threshold"]
        if cen:
            g["fit_mean"], g["fit_std"] = ev2.mean, ev2.std
        t2, n2 = sc2[np.array([t[0] for t in trials]) == 1], sc2[np.array([t[0] for t in trials]) == 0]
        print(f"  center={cen} length_norm={ln}: eer {r2['eer']:.5f} mdc {r2['mdc']:.4f}; target {t2.mean():.4f}+-{t2.std():.4f} "
              f"non-target {n2.mean():.4f}+-{n2.std():.4f}")
    import random as _random
    frames = [30, 50, 75, 149, 149, 10]                             # below / at / above the 50-frame subsample
    rng = np.random.default_rng(99)
    base = rng.standard_normal((3, 64)).astype(np.float32)          # three "speakers": frames scatter round a direction
    np_emb = [torch.from_numpy((base[i % 3] + 0.7 * rng.standard_normal((f, 64))).astype(np.float32))
              for i, f in enumerate(frames)]
    np_samples = [EmbeddingSample(f"np{i}", e) for i, e in enumerate(np_emb)]
    np_trials = [(int(i % 3 == j % 3), i, j) for i in range(6) for j in range(i + 1, 6)]
    _random.seed(1234)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        np_scores = CosineDistanceEvaluator(True, True, 0)._compute_prediction_scores(   # flags are ignored on this branch
            [(np_samples[i], np_samples[j]) for _, i, j in np_trials])
    for i, e in enumerate(np_emb):
        g[f"nonpooled_emb{i}"] = e
    g["nonpooled_trials"] = np.array(np_trials, dtype=np.int64)
    g["nonpooled_scores"] = np.array(np_scores, dtype=np.float64)
    g["nonpooled_seed"] = np.int64(1234)
    if mix is None:
        np.savez_compressed(os.path.join(OUT, "g12_eer.npz"), **to_np(g))
    tgt = scores[np.array([t[0] for t in trials]) == 1]
    non = scores[np.array([t[0] for t in trials]) == 0]
    print(f"g12_eer (mix {kw['mix']}): eer {res['eer']:.5f} mdc {res['mdc']:.4f}; target scores {tgt.mean():.4f}+-{tgt.std():.4f} "
          f"non-target {non.mean():.4f}+-{non.std():.4f}")
    return res


# ----------------------------------------------------------------------------- G13 evaluation-length utterance, G14 third weight seed
def golden_long():
    """the reference tests on whole utterances at batch size 1 (ref: src/main.py:506-514,
    speaker_recognition_module.py:462-500).  One 20 s utterance (N = 320000 -> T = 999) through the reference wrapper,
    eval mode, the g2_base weights: embedding + a strided sample of the hidden states."""
    cfg = O.OracleConfig.base()
    w, _ = build_reference_wrapper(cfg, seed=20211)
    wav, _ = O.synth_batch(1, 320000, 5994, seed=90017)
    w.eval()
    with torch.no_grad():
        h = w(wav[:, 0, :]).transpose(2, 1)
        g = {"eval.mean+std": MeanStdStatPool1D(1)(h), "eval.last_hidden.sample": h[:, ::37, ::16].contiguous()}
    np.savez_compressed(os.path.join(OUT, "g13_long.npz"), **to_np(g))
    print("g13_long: emb norm", float(g["eval.mean+std"].norm()), "T", h.shape[1])


def golden_seed3():
    """a THIRD weight seed for the fp16 embedding bound (the margin under 1e-3 is a few percent, and
    every golden so far used seeds 20211 / 777): weights 4099, utterances 60611, B = 6, 4 s clips (T = 199)."""
    cfg = O.OracleConfig.base()
    w, _ = build_reference_wrapper(cfg, seed=4099)
    wav, _ = O.synth_batch(6, 64000, 5994, seed=60611)
    w.eval()
    with torch.no_grad():
        h = w(torch.squeeze(wav)).transpose(2, 1)
        g = {"eval.mean+std": MeanStdStatPool1D(1)(h), "eval.last_hidden.sample": h[:, ::16, ::16].contiguous()}
    np.savez_compressed(os.path.join(OUT, "g14_seed3.npz"), **to_np(g))
    print("g14_seed3: emb norm", float(g["eval.mean+std"].norm()), "T", h.shape[1])


def golden_seeds():
    """How wide is the fp16 margin across WEIGHT seeds (every golden so far used three seeds)?  Five
    more weight seeds x 4 utterances of 3 s through the reference wrapper (eval, mean+std): g16_seeds.npz holds the five
    [4, 1536] embedding blocks; weights / utterances regenerate from (seed, 7000 + seed)."""
    cfg = O.OracleConfig.base()
    g = {"seeds": np.array([101, 202, 303, 404, 505], dtype=np.int64)}
    for sd_ in g["seeds"].tolist():
        w, _ = build_reference_wrapper(cfg, seed=sd_)
        wav, _ = O.synth_batch(4, 48000, 5994, seed=7000 + sd_)
        w.eval()
        with torch.no_grad():
            g[f"eval.mean+std.{sd_}"] = MeanStdStatPool1D(1)(w(torch.squeeze(wav)).transpose(2, 1))
        print("g16_seeds: seed", sd_, "emb norm", float(g[f"eval.mean+std.{sd_}"].norm()))
    np.savez_compressed(os.path.join(OUT, "g16_seeds.npz"), **to_np(g))

# ----------------------------------------------------------------------------- G17 heavy-tailed weights, G18 the wrapper's "large" branch
def golden_outlier():
    """one base-size weight seed from a HEAVY-TAILED family (w2v2_speaker_amd/data/synthetic.py
    outlier_family: six residual channels with every encoder LayerNorm gain x 20, four FFN-1 bias entries + 8 per block,
    convolutions 1 / 3 / 5 x 3) through the reference wrapper, eval, mean+std: 4 utterances of 3 s.  Also stored: how
    heavy the tail is (max |hidden| over RMS) so the consuming test can say what it was bounded on."""
    cfg = O.OracleConfig.base()
    w, _ = build_reference_wrapper(cfg, seed=20211, family="outlier")
    wav, _ = O.synth_batch(4, 48000, 5994, seed=171717)
    w.eval()
    with torch.no_grad():
        out = w.model(torch.squeeze(wav), output_hidden_states=True)
        h = out.last_hidden_state
        g = {"eval.mean+std": MeanStdStatPool1D(1)(h), "eval.last_hidden.sample": h[:, ::16, ::16].contiguous(),
             "hidden_absmax_over_rms": np.array([float(x.abs().max() / x.pow(2).mean().sqrt()) for x in out.hidden_states]),
             "conv_out_absmax": np.float64(float(out.extract_features.abs().max()))}
        # context for the 16-bit bound: the REFERENCE's own mixed precision (``precision: 16`` of the paper runs = torch
        # autocast; ref config/experiment/speaker_wav2vec2_aam.yaml:17) against its own f32, same weights and utterances,
        # for this family and for the Gaussian family of g2 (torch CPU autocast: 16-bit operands, f32 accumulation)
        e32 = g["eval.mean+std"]
        for dt, tag in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
            with torch.autocast("cpu", dtype=dt):
                e16 = MeanStdStatPool1D(1)(w.model(torch.squeeze(wav)).last_hidden_state.float())
            g[f"ref_autocast_{tag}.per_utt_err"] = (e16.float() - e32).norm(dim=1) / e32.norm(dim=1)
        wg, _ = build_reference_wrapper(cfg, seed=20211)
        wg.eval()
        eg = MeanStdStatPool1D(1)(wg.model(torch.squeeze(wav)).last_hidden_state)
        with torch.autocast("cpu", dtype=torch.float16):
            eg16 = MeanStdStatPool1D(1)(wg.model(torch.squeeze(wav)).last_hidden_state.float())
        g["ref_autocast_fp16.per_utt_err.gaussian_family"] = (eg16.float() - eg).norm(dim=1) / eg.norm(dim=1)
    np.savez_compressed(os.path.join(OUT, "g17_outlier.npz"), **to_np(g))
    print("   reference autocast vs its own f32, per-utterance max: fp16", float(g["ref_autocast_fp16.per_utt_err"].max()),
          "bf16", float(g["ref_autocast_bf16.per_utt_err"].max()), "| Gaussian family fp16",
          float(g["ref_autocast_fp16.per_utt_err.gaussian_family"].max()))
    print("g17_outlier: emb norm", float(g["eval.mean+std"].norm()), "max|h|/rms per layer",
          np.round(g["hidden_absmax_over_rms"], 1), "conv out max", float(g["conv_out_absmax"]))


def golden_large2():
    """the reference wrapper's "large" branch (ref src/models/wav2vec2.py:115-116: num_features 1024) on
    a 2-layer cut of the wav2vec2-large geometry (H 1024, 16 heads, FFN 4096; 5 s clips, T = 249), replacing the
    oracle-only comparison of the HIP path: eval embedding, and -- train mode, no regularisation -- AAM loss (C = 211)
    and the norm of every gradient."""
    import dataclasses
    cfg = dataclasses.replace(O.OracleConfig.large(), num_hidden_layers=2)
    B, N, C = 2, 80000, 211
    w, _ = build_reference_wrapper(cfg, seed=20211, hf_id="facebook/wav2vec2-large")
    assert w.num_features == 1024 and w.model.config.hidden_size == 1024 and w.model.config.num_attention_heads == 16
    wav, label = O.synth_batch(B, N, C, seed=77)
    x = torch.squeeze(wav)
    g = {"label": label}
    w.eval()
    with torch.no_grad():
        h = w(x).transpose(2, 1)
        g["eval.mean+std"] = MeanStdStatPool1D(1)(h)
        g["eval.last_hidden.sample"] = h[:, ::16, ::16].contiguous()
    np.random.seed(11)                                      # injected SpecAugment mask, as g2_base does (seed 7 there)
    mask_t = torch.from_numpy(hf_mod._compute_mask_indices((B, h.shape[1]), mask_prob=0.05, mask_length=10, min_masks=2))
    g["mask"] = mask_t
    w.train()
    out = w.model(x, mask_time_indices=mask_t).last_hidden_state
    emb = MeanStdStatPool1D(1)(out)
    loss_fn = AngularAdditiveMarginSoftMaxLoss(2048, C, margin=0.2, scale=30)
    with torch.no_grad():
        loss_fn.fc_weights.copy_(O.synth_tensor("loss_fn.fc_weights", (C, 2048), 20211))
    loss, pred = loss_fn(emb, label)
    loss.backward()
    g["train.embedding"], g["train.loss"] = emb, loss
    names, norms = [], []
    for n, p in list(w.model.named_parameters()) + [("loss_fn.fc_weights", loss_fn.fc_weights)]:
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(n)
        norms.append(float(gr.double().norm()))
    g["grad_names"], g["grad_norms"] = np.array(names), np.array(norms)
    np.savez_compressed(os.path.join(OUT, "g18_large2.npz"), **to_np(g))
    print("g18_large2: loss", float(loss), "emb norm", float(emb.norm()), "T", h.shape[1])


# ----------------------------------------------------------------------------- G19 pre-LN / layer-norm-conv family
def golden_tiny_stable():
    """VERDICT r5 item 6 / SURVEY App. A.12: the "-lv60" / xlsr family -- HF config flags do_stable_layer_norm=True (pre-LN
    encoder, HF:611-654,729-802), feat_extract_norm="layer" (a LayerNorm after EVERY convolution, HF:275-299), conv_bias=True --
    through the reference wrapper at the tiny geometry with THREE layers: per-stage activations (train mode, injected mask,
    dropouts 0), AAM loss, every gradient; the same with layer 1 skipped (what LayerDrop does: the block is removed from
    the reference's own ModuleList for that pass); eval-mode embedding."""
    import dataclasses
    cfg = dataclasses.replace(O.OracleConfig.tiny(), num_hidden_layers=3, do_stable_layer_norm=True,
                              feat_extract_norm="layer", conv_bias=True)
    B, N, C = 2, 4000, 10
    w, sd = build_reference_wrapper(cfg, seed=20211, hf_id="facebook/wav2vec2-large-lv60")
    assert w.model.config.do_stable_layer_norm and w.model.config.feat_extract_norm == "layer" and w.model.config.conv_bias
    assert type(w.model.encoder).__name__ == "Wav2Vec2EncoderStableLayerNorm"
    wav, label = O.synth_batch(B, N, C, seed=42133724)
    T = cfg.num_frames(N)
    rng = np.random.Generator(np.random.PCG64(7))
    mask = np.zeros((B, T), dtype=bool)
    for b in range(B):
        s0 = rng.integers(0, T - 3)
        mask[b, s0:s0 + 3] = True
    mask_t = torch.from_numpy(mask)
    m = w.model
    aam_w = O.synth_tensor("loss_fn.fc_weights", (C, 2 * cfg.hidden_size), 20211)
    x = torch.squeeze(wav)
    g = {"wav": wav, "label": label, "mask": mask_t}

    def run(tag, grab_stages):
        w.train()
        m.zero_grad()
        stages, hooks = {}, []
        if grab_stages:
            def grab(name):
                def f(_m, _i, o):
                    stages[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
                return f
            hooks.append(m.feature_extractor.register_forward_hook(grab("conv_out_bct")))
            hooks.append(m.feature_projection.register_forward_hook(grab("proj")))
            hooks.append(m.encoder.pos_conv_embed.register_forward_hook(grab("pos_conv")))
            hooks.append(m.encoder.layers[0].register_forward_pre_hook(
                lambda _m, i: stages.__setitem__("enc_in", i[0].detach().clone())))
            for l, layer in enumerate(m.encoder.layers):
                hooks.append(layer.register_forward_hook(grab(f"layer{l}")))
        out = m(x, mask_time_indices=mask_t).last_hidden_state
        emb = MeanStdStatPool1D(dim_to_reduce=1)(out)
        loss_fn = AngularAdditiveMarginSoftMaxLoss(2 * cfg.hidden_size, C, margin=0.2, scale=30)
        with torch.no_grad():
            loss_fn.fc_weights.copy_(aam_w)
        loss, pred = loss_fn(emb, label)
        loss.backward()
        for h in hooks:
            h.remove()
        g[tag + "embedding"], g[tag + "loss"], g[tag + "last_hidden"] = emb, loss, out
        if grab_stages:
            g["stage.conv_out"] = stages["conv_out_bct"].transpose(1, 2)
            for k in ("proj", "pos_conv", "enc_in"):
                g["stage." + k] = stages[k]
            for l in range(cfg.num_hidden_layers):
                g[f"stage.layer{l}"] = stages[f"layer{l}"]
        return loss_fn

    lf = run("", True)
    for n, p in m.named_parameters():
        g["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone()
    g["grad.loss_fn.fc_weights"] = lf.fc_weights.grad.clone()
    # LayerDrop of layer 1: the reference's own modules with that block taken out of the list for one pass
    layers = m.encoder.layers
    full = list(layers)
    m.encoder.layers = torch.nn.ModuleList([full[0], full[2]])
    lf = run("skip1.", False)
    m.encoder.layers = torch.nn.ModuleList(full)
    named = dict(m.named_parameters())
    for n, p in named.items():
        used = not n.startswith("encoder.layers.1.")
        g["skip1.grad." + n] = (p.grad.clone() if (used and p.grad is not None) else torch.zeros_like(p))
    g["skip1.grad.loss_fn.fc_weights"] = lf.fc_weights.grad.clone()
    w.eval()
    with torch.no_grad():
        h = w(x).transpose(2, 1)
        g["eval.last_hidden"], g["eval.mean+std"] = h, MeanStdStatPool1D(1)(h)
    np.savez_compressed(os.path.join(OUT, "g19_tiny_stable.npz"), **to_np(g))
    print("g19_tiny_stable: loss", float(g["loss"]), "skip1 loss", float(g["skip1.loss"]), "T", T, "keys", len(g))


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "base", "aam", "pool", "eval", "optim", "bce", "base2", "base66", "eer", "long", "seed3", "seeds", "outlier", "large2", "tiny_stable"]
    for wname in which:
        {"tiny": golden_tiny, "base": golden_base, "aam": golden_aam, "base66": golden_base66, "pool": golden_pool,
         "eval": golden_eval, "optim": golden_optim, "bce": golden_bce, "base2": golden_base2, "eer": golden_eer,
         "long": golden_long, "seed3": golden_seed3, "seeds": golden_seeds, "outlier": golden_outlier,
         "large2": golden_large2, "tiny_stable": golden_tiny_stable}[wname]()
