"""GPU tests of the reference-shaped module surface (models/, layers/, optim/loss/, lightning_modules/)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import w2v2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_aam_module_matches_reference_known_answers():
    from w2v2_speaker_amd.optim.loss import AngularAdditiveMarginSoftMaxLoss
    g = np.load(os.path.join(GOLDEN, "g4_aam.npz"))
    for margin, scale in ((0.2, 30.0), (0.3, 15.0)):
        k = f"m{margin}_s{scale}."
        fn = AngularAdditiveMarginSoftMaxLoss(24, 7, margin=margin, scale=scale, device=DEV, act_dtype=torch.float32)
        assert (fn.margin, fn.scale, tuple(fn.fc_weights.shape)) == (margin, scale, (7, 24))
        with torch.no_grad():
            fn.fc_weights.copy_(T(g["W"]).to(DEV))
        x = T(g["x"]).to(DEV).requires_grad_(True)
        loss, pred = fn(x, T(g["label"]).to(DEV))
        loss.backward()
        assert abs(float(loss) - float(g[k + "loss"])) < 2e-5
        assert np.allclose(pred.cpu().numpy(), g[k + "softmax"], atol=2e-6)
        # rows 0 / 1 sit on cos == -1 / +1 exactly (theta = pi / 0): d(theta)/dx has no defined direction there;
        # the reference's autograd returns NaN (row 0) or a cancellation residue (row 1), the HIP head a finite
        # vector of the limiting magnitude.  Every regular row must match.
        assert np.allclose(x.grad.cpu().numpy()[2:], g[k + "dx"][2:], atol=2e-5)
        assert torch.isfinite(x.grad).all() and torch.isfinite(fn.fc_weights.grad).all()


def test_aam_module_easy_margin_matches_reference():
    """`easy_margin=True` (ref: src/optim/loss/aam_softmax.py:60-61) through the module mirror -> csrc/heads.hip, against
    the reference's own outputs (tests/golden/g4_aam.npz `easy_*`): loss, softmax, dx of every regular row, dW."""
    from w2v2_speaker_amd.optim.loss import AngularAdditiveMarginSoftMaxLoss
    g = np.load(os.path.join(GOLDEN, "g4_aam.npz"))
    k = "easy_m0.2_s30.0."
    fn = AngularAdditiveMarginSoftMaxLoss(24, 7, margin=0.2, scale=30.0, easy_margin=True, device=DEV, act_dtype=torch.float32)
    with torch.no_grad():
        fn.fc_weights.copy_(T(g["W"]).to(DEV))
    x = T(g["x"]).to(DEV).requires_grad_(True)
    loss, pred = fn(x, T(g["label"]).to(DEV))
    loss.backward()
    assert abs(float(loss) - float(g[k + "loss"])) < 2e-5
    assert np.allclose(pred.cpu().numpy(), g[k + "softmax"], atol=2e-6)
    ok = np.isfinite(g[k + "dx"]).all(axis=1)
    ok[:2] = False                           # cos == -1 / +1 exactly: no defined direction (see the test above)
    assert ok.sum() >= 6 and np.allclose(x.grad.cpu().numpy()[ok], g[k + "dx"][ok], atol=2e-5)
    # dW: the rows 0 / 1 singularities contribute to it in the reference (NaN or residue) -- compare on a batch without them
    x2 = T(g["x"])[2:].to(DEV).requires_grad_(True)
    xr = T(g["x"])[2:].clone().requires_grad_(True)
    Wr = T(g["W"]).clone().requires_grad_(True)
    fn.fc_weights.grad = None
    l2, _ = fn(x2, T(g["label"])[2:].to(DEV))
    l2.backward()
    from oracle import w2v2_oracle as O
    lo, _ = O.aam_softmax(xr, Wr, T(g["label"])[2:], 0.2, 30.0, easy_margin=True)
    lo.backward()
    assert abs(float(l2) - float(lo)) < 2e-5
    assert np.allclose(fn.fc_weights.grad.cpu().numpy(), Wr.grad.numpy(), atol=2e-5)
    assert np.allclose(x2.grad.cpu().numpy(), xr.grad.numpy(), atol=2e-5)


def test_pooling_and_ce_modules_autograd():
    from w2v2_speaker_amd.layers.pooling import IndexPool1D, MaxPool1D, MeanStatPool1D, MeanStdStatPool1D
    from w2v2_speaker_amd.optim.loss import CrossEntropyLoss
    x = torch.randn(3, 21, 64, generator=torch.Generator().manual_seed(1))
    for mod, ref in ((MeanStdStatPool1D(1), lambda t: torch.cat(torch.std_mean(t, 1), 1)),
                     (MeanStatPool1D(1), lambda t: t.mean(1)), (MaxPool1D(1), lambda t: t.max(1).values),
                     (IndexPool1D("first", 1), lambda t: t[:, 0]), (IndexPool1D("middle", 1), lambda t: t[:, -1])):
        xd = x.to(DEV).requires_grad_(True)
        xr = x.double().requires_grad_(True)
        y, yr = mod(xd), ref(xr)
        up = torch.randn(*yr.shape, generator=torch.Generator().manual_seed(2))
        (y * up.to(DEV)).sum().backward()
        (yr * up.double()).sum().backward()
        assert rel_l2(y.detach().cpu(), yr.detach()) < 1e-5 and rel_l2(xd.grad.cpu(), xr.grad) < 1e-5
    # channels-first input (dim_to_reduce=2), as some reference call sites use
    y2 = MeanStdStatPool1D(2)(x.transpose(1, 2).contiguous().to(DEV))
    assert rel_l2(y2.cpu(), torch.cat(torch.std_mean(x, 1), 1)) < 1e-5
    logits = torch.randn(5, 13, generator=torch.Generator().manual_seed(3))
    label = torch.tensor([0, 3, 12, 7, 7])
    ld = logits.to(DEV).requires_grad_(True)
    loss, sm = CrossEntropyLoss()(ld, label.to(DEV))
    loss.backward()
    lr = logits.double().requires_grad_(True)
    lref = torch.nn.functional.cross_entropy(lr, label)
    lref.backward()
    assert abs(float(loss) - float(lref)) < 1e-5 and rel_l2(ld.grad.cpu(), lr.grad) < 1e-5
    assert rel_l2(sm.cpu(), torch.softmax(logits.double(), 1)) < 1e-5


@pytest.mark.parametrize("dim_to_reduce", [1, 2])
def test_attentive_stat_pool_module_vs_restated_definition(dim_to_reduce):
    """ref: src/layers/pooling.py:87-106 ``AttentiveStatPool1D(embedding_size, dim_to_reduce)`` as a class: forward,
    input gradient and all six parameter gradients against autograd over oracle.attentive_stat_pool (speechbrain's
    published definition restated -- speechbrain itself is not installable here, so this is a self-consistency check,
    not reference parity), then eval mode on the running statistics the training call updated."""
    from w2v2_speaker_amd.layers.pooling import AttentiveStatPool1D
    B, Tn, C = 3, 37, 64
    mod = AttentiveStatPool1D(C, dim_to_reduce, device=DEV, act_dtype=torch.float32, init_seed=3)
    views = mod.named_views()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, v in views.items():
            if "running" not in n:
                v.copy_((torch.randn(v.shape, generator=g) * (0.3 if n.endswith("weight") else 0.1)
                         + (1.0 if n.endswith("norm.norm.weight") else 0.0)).to(DEV))
    od = {"tdnn.conv.weight": views["pooling_layer.tdnn.conv.conv.weight"], "tdnn.conv.bias": views["pooling_layer.tdnn.conv.conv.bias"],
          "tdnn.norm.weight": views["pooling_layer.tdnn.norm.norm.weight"], "tdnn.norm.bias": views["pooling_layer.tdnn.norm.norm.bias"],
          "conv.weight": views["pooling_layer.conv.conv.weight"], "conv.bias": views["pooling_layer.conv.conv.bias"]}
    od = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in od.items()}
    x = torch.randn(B, Tn, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = O.attentive_stat_pool(xr, od)
    up = torch.randn(B, 2 * C, generator=g)
    (ref * up).sum().backward()
    xin = (x if dim_to_reduce == 1 else x.transpose(1, 2).contiguous()).to(DEV).requires_grad_(True)
    mod.train()
    out = mod(xin)
    (out * up.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert out.shape == (B, 2 * C) and rel_l2(out.detach().cpu(), ref.detach()) < 2e-5
    gx = xin.grad.cpu() if dim_to_reduce == 1 else xin.grad.cpu().transpose(1, 2)
    assert rel_l2(gx, xr.grad) < 2e-3
    st = mod._store
    flat_grad = mod.flat.grad
    for sb, oname in (("tdnn.conv.conv.weight", "tdnn.conv.weight"), ("tdnn.conv.conv.bias", "tdnn.conv.bias"),
                      ("tdnn.norm.norm.weight", "tdnn.norm.weight"), ("tdnn.norm.norm.bias", "tdnn.norm.bias"),
                      ("conv.conv.weight", "conv.weight"), ("conv.conv.bias", "conv.bias")):
        n = "stat_pooling.pooling_layer." + sb
        o, shp = st.offsets[n], st.shapes[n]
        got = flat_grad[o:o + od[oname].numel()].view(*shp).cpu()
        ref_g = od[oname].grad.view(*shp)
        # (conv.conv.bias: the softmax over time is invariant to a per-channel shift -> its gradient is pure round-off)
        assert float((got - ref_g).norm()) < 2e-3 * float(ref_g.norm()) + 1e-5, sb
    assert float(views["pooling_layer.tdnn.norm.norm.running_mean"].abs().sum()) > 0         # updated by train()
    mod.eval()
    with torch.no_grad():
        e = mod(xin.detach())
    assert e.shape == (B, 2 * C) and torch.isfinite(e).all()


def test_wrapper_module_forward_backward_matches_oracle():
    from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.models.wav2vec2 import Wav2Vec2WrapperModule
    import w2v2_speaker_amd.models.wav2vec2 as wm
    from w2v2_speaker_amd.config import W2V2Config
    cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    orig = W2V2Config.from_huggingface_id
    W2V2Config.from_huggingface_id = staticmethod(lambda _id: cfg)        # tiny shapes behind the "base" id
    try:
        reg = Wav2Vec2RegularisationConfig(attention_dropout=0, feat_proj_dropout=0, hidden_dropout=0, layerdrop=0,
                                           mask_time_prob=0)
        w = Wav2Vec2WrapperModule("facebook/wav2vec2-base", False, reg, device=DEV, act_dtype=torch.float32)
    finally:
        W2V2Config.from_huggingface_id = orig
    sd = O.make_state_dict(ocfg, 20211)
    w.load_state_dict({"model." + k: v for k, v in sd.items()})
    assert set(w.state_dict()) == {"model." + k for k in sd}
    wav, _ = O.synth_batch(2, 4000, 10, seed=5)
    x = wav[:, 0].to(DEV)
    w.eval()
    with torch.no_grad():
        out = w(x)
    ref = O.wav2vec2_forward(wav[:, 0], sd, ocfg)
    assert out.shape == (2, cfg.hidden_size, ref.shape[1])           # [B, num_features, num_frames]
    assert rel_l2(out.transpose(1, 2).cpu(), ref) < 1e-4
    # training mode: gradient wrt the hidden states flows through the hand-written backward
    w.train()
    w.store.zero_grad()
    out = w(x)
    up = torch.randn(*out.shape, generator=torch.Generator().manual_seed(7))
    (out * up.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.wav2vec2_forward(wav[:, 0], osd, ocfg).transpose(1, 2) * up).sum().backward()
    for name in ("encoder.layers.1.feed_forward.output_dense.weight", "encoder.layers.0.attention.q_proj.weight",
                 "feature_projection.projection.weight", "encoder.pos_conv_embed.conv.parametrizations.weight.original1"):
        assert rel_l2(w.store.mg(name).cpu(), osd[name].grad) < 2e-3, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_lite_wrapper_module_conv_features_forward_backward_matches_oracle(dtype):
    """``Wav2vecLiteWrapperModule`` (ref: src/models/wav2vec2.py:149-169, the ``wav2vec_feature_encoder_only`` wrapper):
    forward = the conv feature extractor [B, 512-like, frames], backward = every conv / GroupNorm gradient, against the
    oracle's ``feature_extractor`` (HF:382-419) and its autograd."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.models.wav2vec2 import Wav2vecLiteWrapperModule
    cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    orig = W2V2Config.from_huggingface_id
    W2V2Config.from_huggingface_id = staticmethod(lambda _id: cfg)
    try:
        w = Wav2vecLiteWrapperModule("facebook/wav2vec2-base", True, device=DEV, act_dtype=dtype)
    finally:
        W2V2Config.from_huggingface_id = orig
    sd = O.make_state_dict(ocfg, 20211)
    w.store.load_state_dict({"wav2vec.model." + k: v for k, v in sd.items()}, strict=False)
    assert w.num_embedding_features == cfg.conv_dim[-1]
    wav, _ = O.synth_batch(2, 4000, 10, seed=5)
    x = wav[:, 0].to(DEV)
    ref = O.feature_extractor(wav[:, 0], sd, ocfg)                     # [B, C, T]
    f32 = dtype == torch.float32
    w.eval()
    with torch.no_grad():
        out = w(x)
    assert out.shape == ref.shape and rel_l2(out.float().cpu(), ref) < (1e-5 if f32 else 2e-3)
    w.train()
    w.store.zero_grad()
    out = w(x)
    up = torch.randn(*ref.shape, generator=torch.Generator().manual_seed(9)) * (1.0 if f32 else 64.0)
    (out.float() * up.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.feature_extractor(wav[:, 0], osd, ocfg) * up).sum().backward()
    for i in range(len(cfg.conv_dim)):
        name = f"feature_extractor.conv_layers.{i}.conv.weight"
        assert rel_l2(w.store.mg(name).cpu(), osd[name].grad) < (2e-3 if f32 else 3e-2), name
    for leaf in ("weight", "bias"):
        name = f"feature_extractor.conv_layers.0.layer_norm.{leaf}"
        assert rel_l2(w.store.mg(name).cpu(), osd[name].grad) < (2e-3 if f32 else 3e-2), name
    assert float(w.store.mg("encoder.layers.0.attention.q_proj.weight").abs().max()) == 0.0   # nothing else was touched


def test_fc_module_training_and_eer_parity_on_synthetic_trials():
    """EER on a fixed synthetic trial list from HIP embeddings == EER from the
    oracle's embeddings (f32 parity mode; tiny config for CPU-oracle speed)."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.evaluation.speaker.cosine_distance import EmbeddingSample, EvaluationPair
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import (SpeakerClassificationDataBatch,
                                                                         Wav2vec2FCModule, Wav2vec2FCModuleConfig)
    cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    orig = W2V2Config.from_huggingface_id
    W2V2Config.from_huggingface_id = staticmethod(lambda _id: cfg)
    try:
        mod = Wav2vec2FCModule.from_config(Wav2vec2FCModuleConfig(reset_weights=True), num_speakers=6, device=DEV,
                                           act_dtype=torch.float32, max_lr=1e-3, max_steps=50)
    finally:
        W2V2Config.from_huggingface_id = orig
    sd = O.make_state_dict(ocfg, 20211)
    full = {"wav2vec.model." + k: v for k, v in sd.items()}
    full["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (6, 2 * cfg.hidden_size), 20211)
    mod.load_state_dict(full)
    assert set(mod.state_dict()) == set(full)
    g = np.random.Generator(np.random.PCG64(11))
    base = g.standard_normal((6, 4000)).astype(np.float32)
    wavs, keys, spk = [], [], []
    for s in range(6):
        for u in range(4):
            wavs.append(O.normalise_waveform(torch.from_numpy(base[s] + 0.7 * g.standard_normal(4000).astype(np.float32))))
            keys.append(f"s{s}/u{u}")
            spk.append(s)
    mod.eval()
    outs, oemb = [], {}
    for k, wv in zip(keys, wavs):
        b = SpeakerClassificationDataBatch(1, [k], wv[None, None, :], torch.tensor([0]))
        outs.append(mod.test_step(b))
        oemb[k] = O.speaker_embedding(wv[None, None, :], sd, ocfg, "mean+std")[0]
    pairs = [EvaluationPair(spk[i] == spk[j], keys[i], keys[j]) for i in range(24) for j in range(i + 1, 24)]
    mod.test_pairs = pairs
    res = mod.test_epoch_end(outs)
    ref = mod.evaluator.evaluate(pairs, [EmbeddingSample(k, v) for k, v in oemb.items()])
    assert abs(res["eer"] - ref["eer"]) < 1e-4 and abs(res["mdc"] - ref["mdc"]) < 1e-4, (res, ref)
    with pytest.raises(ValueError):
        mod.test_step(SpeakerClassificationDataBatch(2, keys[:2], torch.stack(wavs[:2])[:, None], torch.tensor([0, 1])))
    # a few optimisation steps through the module's own training_step reduce the loss
    mod.train()
    batch = SpeakerClassificationDataBatch(24, keys, torch.stack(wavs)[:, None, :], torch.tensor(spk))
    steps = [mod.training_step(batch.to(DEV), i) for i in range(12)]
    losses = [float(o["loss"]) for o in steps]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    # train_acc (ref: speaker_recognition_module.py:296-307) = accuracy of the arg-max prediction
    last = steps[-1]
    assert abs(float(last["train_acc"]) - float((last["prediction"].argmax(1).cpu() == torch.tensor(spk)).float().mean())) < 1e-6
    emb, pred = mod(batch.network_input[:3])
    assert emb.shape == (3, 2 * cfg.hidden_size) and pred.shape == emb.shape


def test_initially_frozen_network_trains_head_only_then_unfreezes():
    """ref: wav2vec2_fc.py:339-361 -- while frozen only loss_fn.fc_weights moves; after num_frozen_steps the
    encoder trains too (the CNN stays frozen throughout)."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import (SpeakerClassificationDataBatch,
                                                                         Wav2vec2FCModule, Wav2vec2FCModuleConfig)
    cfg = W2V2Config.tiny()
    orig = W2V2Config.from_huggingface_id
    W2V2Config.from_huggingface_id = staticmethod(lambda _id: cfg)
    try:
        mod = Wav2vec2FCModule.from_config(Wav2vec2FCModuleConfig(wav2vec_initially_frozen=True, num_frozen_steps=2,
                                                                  reset_weights=True),
                                           num_speakers=5, device=DEV, act_dtype=torch.float32, max_lr=1e-2, max_steps=20)
    finally:
        W2V2Config.from_huggingface_id = orig
    wav, label = O.synth_batch(4, 4000, 5, seed=9)
    batch = SpeakerClassificationDataBatch(4, ["a", "b", "c", "d"], wav, label).to(DEV)
    mod.train()
    mod.on_train_start()
    st = mod.store
    h = st.head_size()
    before = st.flat.clone()
    mod.training_step(batch, 0)
    torch.cuda.synchronize()
    assert not torch.equal(st.flat[:h], before[:h])            # head moved
    assert torch.equal(st.flat[h:], before[h:])                # everything else frozen
    mod.training_step(batch, 1)
    assert mod._is_wav2vec_frozen is False                      # 2 backward calls -> unfrozen
    mid = st.flat.clone()
    mod.training_step(batch, 2)
    torch.cuda.synchronize()
    assert not torch.equal(st.flat[h:st.n_train], mid[h:st.n_train])
    assert torch.equal(st.flat[st.n_train:], before[st.n_train:])   # CNN never updated


def test_reference_construction_sequence_hidden_fc_layers_and_handles():
    """the construction sequence of ref: src/main.py:223-285 (construct_speaker_recognition_module:
    kwargs hyperparameters_to_save / cfg / num_speakers / loss_fn_constructor / validation_pairs / test_pairs /
    evaluator, where loss_fn_constructor = lambda: instantiate(cfg.optim.loss), main.py:296-300) against the mirror,
    with hidden FC layers (ref: wav2vec2_fc.py:185-228) and embedding_layer_idx (ref: :277-288, :363-412), the
    wav2vec.model.* handles the reference touches (:347,361) and strict / non-strict state-dict loading.  The
    FC + CE arithmetic (forward, loss, every head gradient) is checked against torch autograd on the same weights."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.evaluation.speaker.cosine_distance import CosineDistanceEvaluator
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import (SpeakerClassificationDataBatch,
                                                                         Wav2vec2FCModule, Wav2vec2FCModuleConfig)
    from w2v2_speaker_amd.optim.loss import AngularAdditiveMarginSoftMaxLoss, CrossEntropyLoss
    cfg_m, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    H = cfg_m.hidden_size
    orig = W2V2Config.from_huggingface_id
    W2V2Config.from_huggingface_id = staticmethod(lambda _id: cfg_m)
    try:
        network_cfg = Wav2vec2FCModuleConfig(hidden_fc_layers_out=[48, 24], embedding_layer_idx=0, reset_weights=True,
                                             attention_dropout=0.0, feat_proj_dropout=0.0, hidden_dropout=0.0,
                                             layerdrop=0.0, mask_time_prob=0.0)
        kwargs = {"hyperparameters_to_save": {"optim": {"loss": "cross_entropy"}}, "cfg": network_cfg, "num_speakers": 9,
                  "loss_fn_constructor": lambda: CrossEntropyLoss(), "validation_pairs": [], "test_pairs": [],
                  "evaluator": CosineDistanceEvaluator(False, False, 0)}
        net = Wav2vec2FCModule(**kwargs, device=DEV, act_dtype=torch.float32, max_lr=1e-3, max_steps=10)
        aam = Wav2vec2FCModule(**{**kwargs, "cfg": Wav2vec2FCModuleConfig(reset_weights=True),
                                  "loss_fn_constructor": lambda: AngularAdditiveMarginSoftMaxLoss(
                                      2, 2, margin=0.3, scale=15, device=DEV, act_dtype=torch.float32)},
                               device=DEV, act_dtype=torch.float32)
    finally:
        W2V2Config.from_huggingface_id = orig
    assert (aam.loss, aam.margin, aam.scale) == ("aam", 0.3, 15.0) and aam.embedding_size == 2 * H
    assert net.embedding_size == 48 and net.stat_pool_dimension == 2 * H
    assert {"fc_list.0.0.weight", "fc_list.1.0.bias", "fc_list.2.0.weight"} <= set(net.state_dict())
    assert tuple(net.store.shapes["fc_list.2.0.weight"]) == (9, 24)
    # handles of ref: wav2vec2_fc.py:339-361
    net.on_train_start()
    net.wav2vec.model.feature_extractor.requires_grad_(False)            # the default freeze: accepted
    with pytest.raises(RuntimeError):
        net.wav2vec.model.feature_extractor.requires_grad_(True)         # no CNN gradient buffers in this arena
    assert net.wav2vec.num_features == H and len(list(net.wav2vec.model.encoder.parameters())) > 0
    # strict load: missing / unexpected keys raise, non-strict skips them
    sd = net.state_dict()
    with pytest.raises(KeyError):
        net.load_state_dict({k: v for k, v in sd.items() if k != "fc_list.1.0.bias"}, strict=True)
    with pytest.raises(KeyError):
        net.load_state_dict({**sd, "not.a.key": torch.zeros(1)}, strict=True)
    net.load_state_dict({**sd, "not.a.key": torch.zeros(1)}, strict=False)
    # arithmetic: pooled embedding (oracle) -> Linear/ReLU x2 -> Linear -> CE, against autograd
    osd = {k[len("wav2vec.model."):]: v for k, v in sd.items() if k.startswith("wav2vec.model.")}
    wav, label = O.synth_batch(5, 4000, 9, seed=13)
    pooled = O.speaker_embedding(wav, osd, ocfg, "mean+std").detach()
    Ws = [sd[f"fc_list.{i}.0.weight"].clone().requires_grad_(True) for i in range(3)]
    bs = [sd[f"fc_list.{i}.0.bias"].clone().requires_grad_(True) for i in range(3)]
    pr = pooled.clone().requires_grad_(True)
    h0 = torch.relu(pr @ Ws[0].t() + bs[0])
    h1 = torch.relu(h0 @ Ws[1].t() + bs[1])
    logits = h1 @ Ws[2].t() + bs[2]
    loss_ref = torch.nn.functional.cross_entropy(logits, label)
    loss_ref.backward()
    net.eval()
    emb, pred = net(wav.to(DEV))
    assert emb.shape == (5, 48) and rel_l2(emb.cpu(), h0.detach()) < 1e-4          # embedding_layer_idx = 0
    assert rel_l2(pred.cpu(), logits.detach()) < 1e-4
    net.train()
    st = net.store
    before = {n: st.p(n).clone() for n in st.shapes}
    plan = net._plan(5, 4000, True)
    st.zero_grad()
    plan.embed(wav.to(DEV), None, ())
    loss, sm = plan.head_forward_backward(label.to(DEV))
    plan.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 1e-4 and rel_l2(sm.cpu(), torch.softmax(logits.detach(), 1)) < 1e-4
    for i in range(3):
        assert rel_l2(st.g(f"fc_list.{i}.0.weight").cpu(), Ws[i].grad) < 1e-3, i
        assert rel_l2(st.g(f"fc_list.{i}.0.bias").cpu(), bs[i].grad) < 1e-3, i
    assert rel_l2(plan.demb.cpu(), pr.grad) < 1e-3                                 # gradient reaching the pooling
    out = net.training_step(SpeakerClassificationDataBatch(5, list("abcde"), wav, label).to(DEV), 0)
    assert np.isfinite(float(out["loss"])) and 0.0 <= float(out["train_acc"]) <= 1.0
    assert any(not torch.equal(st.p(n), before[n]) for n in ("fc_list.0.0.weight", "fc_list.2.0.bias"))


def test_ensemble_of_layers_embeddings_vs_oracle():
    """ref: wav2vec2_fc.py:440-463: pooled embedding of each of the last n hidden states (HF output_hidden_states)."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.params import ParamStore
    cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    st = ParamStore(cfg, DEV, torch.float32, head=None)
    sd = O.make_state_dict(ocfg, 20211)
    st.load_state_dict(sd)
    wav, _ = O.synth_batch(3, 4000, 10, seed=4)
    plan = Plan(st, 3, 4000, train=False, keep_hidden_states=True)
    embs = plan.ensemble_embeddings(wav.to(DEV), 2)
    torch.cuda.synchronize()
    _, stages = O.wav2vec2_forward(wav[:, 0], sd, ocfg, return_stages=True)
    refs = [O.mean_std_pool(stages["layer0"]), O.mean_std_pool(stages["layer1"])]
    assert len(embs) == 2
    for e, r in zip(embs, refs):
        assert rel_l2(e.cpu(), r) < 2e-5
    hs = plan.hidden_states()
    assert len(hs) == cfg.num_hidden_layers + 1 and rel_l2(hs[-1].cpu(), stages["layer1"]) < 2e-5



def test_training_steps_fed_from_reference_format_shards(tmp_path):
    """shards in the reference's tar format -> normalise -> random 3 s crop -> shuffle-queue batches
    -> pinned-memory side-stream upload (DeviceFeeder) -> training steps; the loss goes down on a 4-speaker toy set."""
    import random
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.data import DeviceFeeder, ShardDataset, write_shards
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import Constant
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    g = torch.Generator().manual_seed(0)
    data = []
    for i in range(32):
        spk = i % 4
        t = torch.arange(52000) / 16000.0
        wav = torch.sin(2 * np.pi * (200.0 + 150.0 * spk) * t)[None] + 0.05 * torch.randn(1, 52000, generator=g)
        data.append((f"id{spk}/yt{i}/{i:05d}", spk, wav))
    paths = write_shards(data, str(tmp_path), samples_per_shard=16)
    cfg = W2V2Config.tiny()
    st = ParamStore(cfg, DEV, torch.float32, head="aam", num_speakers=4)
    st.init_weights(seed=1)
    reg = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                                       hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
    plan = Plan(st, 8, 48000, train=True, reg=reg)
    tr = SpeakerTrainer(st, plan, Constant(2e-3, 0.9))
    random.seed(0)
    losses = []
    for epoch in range(6):
        for batch in DeviceFeeder(ShardDataset(paths, batch_size=8, queue_size=16), DEV):
            assert batch.network_input.is_cuda and batch.network_input.shape == (8, 1, 48000)
            loss, _ = tr.train_step(batch.network_input, batch.ground_truth, skip_layers=())
            losses.append(float(loss))
    assert all(np.isfinite(losses)) and np.mean(losses[-4:]) < 0.7 * np.mean(losses[:4]), losses


def test_ecapa_and_paired_module_surfaces_train():
    """Mirrors of the reference's EcapaTdnnModule and Wav2vec2PairedSpeakerModule: config field names, method names,
    one training step each through the public surface, finite decreasing losses."""
    from w2v2_speaker_amd import config as C
    from w2v2_speaker_amd.lightning_modules.speaker.ecapa_tdnn import EcapaTDNNModuleConfig, EcapaTdnnModule
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import SpeakerClassificationDataBatch
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_paired_input import (
        PairedSpeakerClassificationDataBatch, Wav2vec2PairedSpeakerModule, Wav2vec2PairedSpeakerModuleConfig)
    g = torch.Generator().manual_seed(0)
    ecfg = EcapaTDNNModuleConfig(input_mel_coefficients=16, lin_neurons=24, channels=[64, 64, 64, 64, 192],
                                 attention_channels=16, res2net_scale=4, se_channels=16)
    # the reference's constructor (ref: ecapa_tdnn.py:51-62, what src/main.py:256-285 passes)
    from w2v2_speaker_amd.optim.loss import AngularAdditiveMarginSoftMaxLoss, BinaryCrossEntropyLoss
    actor = lambda: AngularAdditiveMarginSoftMaxLoss(2, 2, margin=0.2, scale=30.0, device=DEV, act_dtype=torch.float32)
    em = EcapaTdnnModule(None, ecfg, 5, actor, [], [], None, max_lr=2e-3, max_steps=50)
    feat = torch.randn(6, 40, 16, generator=g)
    batch = SpeakerClassificationDataBatch(6, [str(i) for i in range(6)], feat, torch.randint(0, 5, (6,), generator=g))
    losses = [float(em.training_step(batch)["loss"]) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    emb, pred = em(feat)
    assert emb.shape == (6, 24) and torch.isfinite(emb).all()
    assert em.generate_example_input(True, 3).shape == (3, 100, 16)
    tiny = C.W2V2Config.tiny()
    orig = C.W2V2Config.from_huggingface_id
    C.W2V2Config.from_huggingface_id = staticmethod(lambda _id: tiny)
    try:
        # ref: wav2vec2_paired_input.py:65-71 (hyperparameters_to_save, cfg, loss_fn_constructor); fp16 by default
        pm = Wav2vec2PairedSpeakerModule(None, Wav2vec2PairedSpeakerModuleConfig(), BinaryCrossEntropyLoss,
                                         max_lr=2e-3, max_steps=50)
        assert pm.store.act_dtype == torch.float16 and pm.store.scaler is not None
        pm.store.scaler[0] = 256.0
        a, b = 0.3 * torch.randn(4, 4000, generator=g), 0.3 * torch.randn(4, 4000, generator=g)
        pb = PairedSpeakerClassificationDataBatch(4, list("abcd"), a, list("efgh"), b, torch.tensor([1, 0, 1, 0]))
        pl = [float(pm.training_step(pb)["loss"]) for _ in range(10)]
        assert np.isfinite(pl).all() and pl[-1] < pl[0]
        scores = pm(a, b)
        assert scores.shape == (4, 1) and torch.isfinite(scores).all()
        # the loss mirror on its own: reference semantics (mean BCE-with-logits, sigmoid prediction), differentiable
        lg = scores.detach().clone().float().requires_grad_(True)
        lab = torch.tensor([1, 0, 1, 0], device=lg.device)
        loss, pred = BinaryCrossEntropyLoss()(lg, lab)
        loss.backward()
        ref = torch.nn.functional.binary_cross_entropy_with_logits(lg.detach()[:, 0], lab.float())
        assert abs(float(loss) - float(ref)) < 1e-5 and torch.allclose(pred, torch.sigmoid(lg.detach()[:, 0]), atol=1e-6)
        assert torch.allclose(lg.grad[:, 0], (torch.sigmoid(lg.detach()[:, 0]) - lab.float()) / 4, atol=1e-6)
    finally:
        C.W2V2Config.from_huggingface_id = orig
