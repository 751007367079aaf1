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
        mod = Wav2vec2FCModule(Wav2vec2FCModuleConfig(), num_speakers=6, device=DEV, act_dtype=torch.float32,
                               max_lr=1e-3, max_steps=50)
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
    losses = [float(mod.training_step(batch.to(DEV), i)["loss"]) for i in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
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
        mod = Wav2vec2FCModule(Wav2vec2FCModuleConfig(wav2vec_initially_frozen=True, num_frozen_steps=2),
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
    em = EcapaTdnnModule(ecfg, num_speakers=5, max_lr=2e-3, max_steps=50)
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
        pm = Wav2vec2PairedSpeakerModule(Wav2vec2PairedSpeakerModuleConfig(), max_lr=2e-3, max_steps=50)
        a, b = 0.3 * torch.randn(4, 4000, generator=g), 0.3 * torch.randn(4, 4000, generator=g)
        pb = PairedSpeakerClassificationDataBatch(4, list("abcd"), a, list("efgh"), b, torch.tensor([1, 0, 1, 0]))
        pl = [float(pm.training_step(pb)["loss"]) for _ in range(10)]
        assert np.isfinite(pl).all() and pl[-1] < pl[0]
        scores = pm(a, b)
        assert scores.shape == (4, 1) and torch.isfinite(scores).all()
    finally:
        C.W2V2Config.from_huggingface_id = orig
