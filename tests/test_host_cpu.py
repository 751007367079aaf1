"""CPU-only tests: the C-ABI library loads and exports every symbol the header declares, host logic
(SpecAugment sampler, one-cycle schedule, flat parameter arena, bucketed all-reduce over gloo)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_exports_every_header_symbol():
    from w2v2_speaker_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "w2v2_hip.h")).read()
    declared = set(re.findall(r"\b(w2v2_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()                      # dlopen works without a GPU
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/w2v2_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.w2v2_version() >= 100


def test_ops_refuse_cpu_tensors():
    from w2v2_speaker_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.colsum(torch.zeros(4, 8), torch.zeros(8), 4, 8)


def test_spec_augment_matches_hf_masks():
    from w2v2_speaker_amd.spec_augment import compute_mask_indices
    g = np.load(os.path.join(GOLDEN, "g8_optim.npz"))
    for seed, shape in ((7, (66, 149)), (11, (4, 249))):
        np.random.seed(seed)
        m = compute_mask_indices(shape, 0.05, 10, min_masks=2)
        assert np.array_equal(m, g[f"mask.seed{seed}"])
        assert m.sum(axis=1).min() >= 10      # >= 1 span survives overlap


def test_one_cycle_matches_torch_table():
    from w2v2_speaker_amd.optim.schedule import OneCycle
    g = np.load(os.path.join(GOLDEN, "g8_optim.npz"))
    s = OneCycle(max_lr=1e-4, total_steps=100)
    for i in range(100):
        lr, b1 = s.at(i)
        assert abs(lr - g["lr"][i]) < 1e-12 and abs(b1 - g["beta1"][i]) < 1e-9
    with pytest.raises(ValueError):
        s.at(100)


def test_param_arena_layout_and_buckets():
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore, W2V_PREFIX
    cfg = W2V2Config.tiny()
    st = ParamStore(cfg, "cpu", torch.float32, head="aam", num_speakers=10)
    # q/k/v adjacent -> fused [3H,H] view aliases the three HF parameters
    w = st.qkv(1, "p")
    st.mp("encoder.layers.1.attention.k_proj.weight").fill_(2.0)
    assert float(w[cfg.hidden_size:2 * cfg.hidden_size].min()) == 2.0 and float(w[:cfg.hidden_size].max()) == 0.0
    # frozen CNN is the tail of the arena and has no gradient
    assert not st.is_trainable(W2V_PREFIX + "feature_extractor.conv_layers.3.conv.weight")
    with pytest.raises(KeyError):
        st.mg("feature_extractor.conv_layers.3.conv.weight")
    # buckets tile the trainable range exactly, in backward order
    b = st.grad_buckets()
    assert b[0][0] == "head" and b[0][1] == 0 and b[-1][2] == st.n_train
    for (n0, s0, e0), (n1, s1, e1) in zip(b, b[1:]):
        assert e0 == s1 and e0 > s0
    assert [n for n, _, _ in b] == ["head", "layer1", "layer0", "prologue", "projection"]
    # the late, unhideable slice is the small one (projection + masked embed + feature LayerNorm), the pos-conv pair
    # rides in the bucket that is final before the pos-conv data gradient runs
    bk = {n: (s, e) for n, s, e in b}
    assert bk["projection"][1] - bk["projection"][0] < bk["prologue"][1] - bk["prologue"][0]
    # state-dict round trip incl. the old weight_g / weight_v naming
    sd = {k: torch.randn(v) for k, v in st.shapes.items()}
    old = {k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v"): v
           for k, v in sd.items()}
    st.load_state_dict(old)
    out = st.state_dict()
    for k in sd:
        assert torch.equal(out[k], sd[k]), k


def test_base_param_count_matches_reference_log():
    """ref: paper_results/auto_lr_find/wav2vec2-sv-aam/run.log:13-25 -- AAM 9.2 M + wav2vec 94.4 M."""
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore
    st = ParamStore(W2V2Config(), "cpu", torch.float32, head="aam", num_speakers=5994)
    n = sum(int(np.prod(s)) for s in st.shapes.values())
    assert n == 94_371_712 + 5994 * 1536
    trainable = sum(int(np.prod(s)) for k, s in st.shapes.items() if st.is_trainable(k))
    assert trainable == 99_378_048                    # SURVEY 2.3 C1 (frozen feature extractor)


def test_ecapa_param_count_matches_reference_log():
    """ref: paper_results/auto_lr_find/ecapa/run.log:14-17 prints ``ECAPA_TDNN 20.8 M``, ``Classifier 232 K``, ``21.0 M
    Trainable params`` for ``n_mels: 80`` (.hydra/config.yaml:88) and the 1211 speakers of VoxCeleb1.  speechbrain is
    not installable here, so this does not pin the ECAPA arithmetic -- it pins the LAYER STRUCTURE of the restatement
    (every conv / norm / SE / pooling tensor shape) against the one number the reference itself recorded, and the
    engine's parameter arena against the restatement."""
    from oracle.ecapa_oracle import EcapaConfig, param_shapes

    def human(n):          # pytorch_lightning.utilities.model_summary.get_human_readable_count (1.4.x)
        for div, unit in ((1e9, "B"), (1e6, "M"), (1e3, "K")):
            if n >= div:
                v = n / div
                return f"{v:.1f} {unit}" if v < 100 and unit != "K" else f"{int(v)} {unit}"
        return str(n)

    shapes = param_shapes(EcapaConfig(input_size=80))
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == 20_767_552 and human(n) == "20.8 M"
    n_cls = 1211 * 192                                        # speechbrain Classifier: cosine weight [speakers, lin_neurons]
    assert human(n_cls) == "232 K" and human(n + n_cls) == "21.0 M"
    assert sum(int(np.prod(s)) for s in param_shapes(EcapaConfig(input_size=40)).values()) == 20_562_752
    from w2v2_speaker_amd.ecapa import FE, EcapaStore
    from w2v2_speaker_amd.ecapa import EcapaConfig as EngineConfig
    st = EcapaStore(EngineConfig(input_mel_coefficients=80), "cpu", torch.float32, num_speakers=1211)
    body = {k: v for k, v in st.shapes.items() if k.startswith(FE)}
    assert sum(int(np.prod(s)) for s in body.values()) == n
    assert {k[len(FE):] for k in body} == set(shapes)          # same speechbrain state-dict names


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import BucketAllReducer
    st = ParamStore(W2V2Config.tiny(), "cpu", torch.float32, head="aam", num_speakers=10)
    red = BucketAllReducer(st, bucket_merge=2)
    gen = torch.Generator().manual_seed(100 + rank)
    st.grad.copy_(torch.randn(st.n_train, generator=gen))
    mine = st.grad.clone()
    for name, _, _ in st.grad_buckets():          # the order backward fires them
        red.bucket_ready(name)
    red.wait()
    others = [torch.randn(st.n_train, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    q.put((rank, bool(torch.allclose(st.grad, sum(others), atol=1e-6)), bool(torch.equal(mine, others[rank]))))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and same for _, ok, same in res), res


def _bcast_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import BucketAllReducer
    st = ParamStore(W2V2Config.tiny(), "cpu", torch.float32, head="aam", num_speakers=10)
    st.init_weights(seed=1000 + rank)                      # every rank starts DIFFERENT ("load on rank 0" situation)
    g = torch.Generator().manual_seed(50 + rank)
    st.exp_avg = torch.randn(st.n_train, generator=g)
    st.exp_avg_sq = torch.rand(st.n_train, generator=g)
    red = BucketAllReducer(st, bucket_merge=2)
    before = st.flat.clone()
    v0 = st.version
    # ONLY rank 0 has "loaded a checkpoint": its Adam step counts and schedule position are advanced (the
    # host half of the optimiser state must travel with the moments, else the other ranks run bias-correction t = 1 and
    # schedule.at(0) on step-N moments)
    if rank == 0:
        st.set_step_counts(41, 37)
    got = red.broadcast_parameters(root=0, host_counters=[1234 if rank == 0 else 0, 77 if rank == 0 else 5])
    host_ok = (st.step_head, st.step_body, st.step_count) == (41, 37, 41) and got == [1234, 77]
    ref = ParamStore(W2V2Config.tiny(), "cpu", torch.float32, head="aam", num_speakers=10)
    ref.init_weights(seed=1000)
    g0 = torch.Generator().manual_seed(50)
    m0 = torch.randn(st.n_train, generator=g0)
    s0 = torch.rand(st.n_train, generator=g0)
    ok = (torch.equal(st.flat, ref.flat) and torch.equal(st.exp_avg, m0) and torch.equal(st.exp_avg_sq, s0)
          and st.version > v0 and (rank == 0 or not torch.equal(before, st.flat)) and host_ok)
    # a rank whose state differs in KIND (no moments) must be refused, not silently mis-paired
    refused = None
    if world == 2:
        if rank == 1:
            st.exp_avg = st.exp_avg_sq = None
        try:
            red.broadcast_parameters(root=0)
            refused = False
        except RuntimeError:
            refused = True
    q.put((rank, bool(ok), refused))
    dist.destroy_process_group()


def test_broadcast_parameters_two_ranks_gloo():
    """SURVEY C2 / ref: config/trainer/trainer.yaml:6-12 -- PL's DDP wrapper broadcasts rank 0's module state at start-up;
    here ``BucketAllReducer.broadcast_parameters``: two ranks with DIFFERENT initial weights and moments end up with rank
    0's, bit for bit; ranks that disagree on which state tensors exist get an error on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and refused for _, ok, refused in res), res


def _check_schedule(L, skip, pair_uppers, members, group=2):
    """Invariants of engine.encoder_backward_schedule against the bucketing of trainer.BucketAllReducer."""
    from w2v2_speaker_amd.engine import encoder_backward_schedule
    ev = encoder_backward_schedule(L, skip, pair_uppers, True, group)
    if group > 2:        # a launch covers at most `group` layers, and only the last launch may cover fewer (groups span skips)
        sizes = [len(e[1]) for e in ev if e[0] == "wgrad"]
        assert all(n == group for n in sizes[:-1]) and all(1 <= n <= group for n in sizes[-1:]), sizes
    last_write, parked, notified = {}, [], []
    for i, e in enumerate(ev):
        if e[0] == "body":
            assert e[1] not in skip
            parked.append(e[1])                       # its LayerNorm partials wait for the next fold
        elif e[0] == "wgrad":
            for l in e[1]:
                assert l not in skip and ("w", l) not in last_write, "weight gradients of a layer are written once"
                last_write[("w", l)] = i
        elif e[0] == "fold":
            for l in parked:
                last_write[("ln", l)] = i
            parked = []
        elif e[0] == "notify":
            notified.append((i, e[1]))
    assert [l for _, l in notified] == list(range(L - 1, -1, -1)), "every layer bucket once, in backward order"
    assert not parked
    when = {l: i for i, l in notified}
    for l in range(L):
        if l in skip:
            assert ("w", l) not in last_write and ("ln", l) not in last_write       # zero gradient, never written
            continue
        assert last_write[("w", l)] < when[l] and last_write[("ln", l)] < when[l], (l, skip)
    # a merged collective fires on its LAST member: every writer of every member must have run by then
    for fire, mem in members.items():
        if not fire.startswith("layer"):
            continue
        t = when[int(fire[5:])]
        for m in mem:
            lm = int(m[5:])
            assert when[lm] <= t
            if lm not in skip:
                assert last_write[("w", lm)] < t and last_write[("ln", lm)] < t, (fire, m, skip)


def test_backward_schedule_notifies_every_bucket_after_its_last_writer():
    """Plan.backward's order of launches and bucket notifications (LayerDrop skip patterns, paired
    weight-gradient launches, deferred LayerNorm folds) against the reducer's merged buckets -- exhaustively for 6
    layers, randomly for the 12- and 24-layer models."""
    import dataclasses
    import itertools
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import BucketAllReducer
    rng = np.random.RandomState(0)
    for L in (1, 2, 5, 6, 12, 24):
        cfg = dataclasses.replace(W2V2Config.tiny(), num_hidden_layers=L)
        st = ParamStore(cfg, "cpu", torch.float32, head="aam", num_speakers=10)
        pair_uppers = list(range(L - 1, 0, -2))                       # Plan._build_gemms
        for merge in (1, 2, 3):
            members = BucketAllReducer(st, bucket_merge=merge).members
            covered = sorted(m for mem in members.values() for m in mem)
            assert covered == sorted(n for n, _, _ in st.grad_buckets())            # each raw bucket in one collective
            if L <= 6:
                pats = [tuple(i for i in range(L) if (mask >> i) & 1) for mask in range(1 << L)]
            else:
                pats = [()] + [tuple(np.nonzero(rng.rand(L) < p)[0].tolist()) for p in (0.05, 0.3, 0.5, 0.9) for _ in range(40)]
            for skip in pats:
                _check_schedule(L, set(skip), pair_uppers, members)
                _check_schedule(L, set(skip), [], members)            # W2V2_NO_WGRAD_PAIRS
                _check_schedule(L, set(skip), pair_uppers, members, group=4)      # wav2vec2-large: groups of four
                _check_schedule(L, set(skip), pair_uppers, members, group=3)


def test_eval_metrics_and_evaluator_match_reference_golden():
    from w2v2_speaker_amd.eval_metrics import calculate_eer, calculate_mdc
    from w2v2_speaker_amd.evaluation.speaker.cosine_distance import (CosineDistanceEvaluator, EmbeddingSample,
                                                                     EvaluationPair, compute_cosine_scores)
    g = np.load(os.path.join(GOLDEN, "g6_eval.npz"))
    eer, thr = calculate_eer(g["gt"].tolist(), g["scores"].tolist())
    assert abs(eer - float(g["eer"])) < 1e-12 and abs(thr - float(g["eer_thr"])) < 1e-12
    mdc, mthr = calculate_mdc(g["gt"].tolist(), g["scores"].tolist())
    assert abs(mdc - float(g["mdc"])) < 1e-12 and abs(mthr - float(g["mdc_thr"])) < 1e-12
    a, b = torch.from_numpy(g["cos_a"]), torch.from_numpy(g["cos_b"])
    assert np.allclose(compute_cosine_scores(a, b), g["cos"], atol=1e-6)
    # evaluator end to end: (s+1)/2 clip, EER/minDCF of the same trials
    samples = [EmbeddingSample(f"a{i}", a[i]) for i in range(50)] + [EmbeddingSample(f"b{i}", b[i]) for i in range(50)]
    pairs = [EvaluationPair(bool(i % 2), f"a{i}", f"b{i}") for i in range(50)]
    res = CosineDistanceEvaluator(False, False, 0).evaluate(pairs, samples)
    ref_eer, _ = calculate_eer([i % 2 for i in range(50)], g["cos01"].tolist())
    assert abs(res["eer"] - ref_eer) < 1e-9
    with pytest.raises(ValueError):
        calculate_eer([0, 2], [0.1, 0.2])
    missing = CosineDistanceEvaluator().evaluate([EvaluationPair(True, "zz", "a0")], samples)
    assert missing["eer"] == -1


def test_ensemble_of_layers_scoring_is_mean_of_member_scores():
    """ref: src/evaluation/speaker/cosine_distance.py:134-185."""
    import torch
    from w2v2_speaker_amd.evaluation.speaker.cosine_distance import CosineDistanceEvaluator, EmbeddingSample
    g = torch.Generator().manual_seed(0)
    n, d = 3, 16
    samples = {k: [torch.randn(d, generator=g) for _ in range(n)] for k in "abcd"}
    pairs = [(EmbeddingSample("a", samples["a"]), EmbeddingSample("b", samples["b"])),
             (EmbeddingSample("c", samples["c"]), EmbeddingSample("d", samples["d"]))]
    ev = CosineDistanceEvaluator(False, False, 0)
    got = ev._compute_prediction_scores(pairs)
    for (l, r), s in zip((("a", "b"), ("c", "d")), got):
        ref = sum(float(torch.nn.functional.cosine_similarity(samples[l][i], samples[r][i], dim=0)) for i in range(n)) / n
        assert abs(s - ref) < 1e-6
    bad = [(EmbeddingSample("a", samples["a"]), EmbeddingSample("b", samples["b"][:2]))]
    try:
        ev._compute_prediction_scores(bad)
        assert False
    except ValueError:
        pass


def test_pl_format_checkpoint_round_trip_and_reheading(tmp_path):
    """state_dict under the reference's key names in a PL-style file; non-strict reload into a
    module with a different number of speakers keeps the network and re-initialises only the head."""
    import torch
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import Wav2vec2FCModule, Wav2vec2FCModuleConfig
    import dataclasses
    from w2v2_speaker_amd import config as C
    from w2v2_speaker_amd.params import ParamStore
    ParamStoreLegacy = ParamStore.legacy_key
    tiny = C.W2V2Config.tiny()
    orig = C.W2V2Config.from_huggingface_id
    C.W2V2Config.from_huggingface_id = staticmethod(lambda _id: tiny)       # keep the CPU test small
    try:
        kw = dict(device="cpu", act_dtype=torch.float32)
        a = Wav2vec2FCModule.from_config(Wav2vec2FCModuleConfig(reset_weights=True), num_speakers=7, init_seed=1, **kw)
        path = str(tmp_path / "last.ckpt")
        a.steps = a.schedule_step = 123
        a.save_checkpoint(path)
        ck = torch.load(path, weights_only=False)
        assert {"state_dict", "global_step", "pytorch-lightning_version"} <= set(ck)
        assert "loss_fn.fc_weights" in ck["state_dict"]
        assert "wav2vec.model.encoder.layers.0.attention.q_proj.weight" in ck["state_dict"]
        from w2v2_speaker_amd.optim.loss import AngularAdditiveMarginSoftMaxLoss
        ctor = lambda: AngularAdditiveMarginSoftMaxLoss(2, 2, margin=0.2, scale=30, device="cpu", act_dtype=torch.float32)
        ref_kw = dict(hyperparameters_to_save=None, num_speakers=7, loss_fn_constructor=ctor, validation_pairs=[],
                      test_pairs=[], evaluator=None)          # what ref: src/main.py:256-283 passes
        # the file must carry the names of the reference's own stack (torch 1.9 weight_norm) -- its
        # strict=False load would silently drop the parametrization names -- and torch-Adam-shaped optimizer states
        keys = set(ck["state_dict"])
        assert "wav2vec.model.encoder.pos_conv_embed.conv.weight_g" in keys
        assert "wav2vec.model.encoder.pos_conv_embed.conv.weight_v" in keys
        assert not any("parametrizations" in k for k in keys)
        assert ck["pytorch-lightning_version"] == "1.4.5"
        osd = ck["optimizer_states"][0]
        assert set(osd) == {"state", "param_groups"} and osd["param_groups"][0]["params"] == list(range(len(keys)))
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(tuple(v.shape))) for v in
                                (ck["state_dict"][ParamStoreLegacy(k)] for k in a.store.reference_parameter_order())])
        opt.load_state_dict(osd)                # what PL's restore does with it
        b = Wav2vec2FCModule.load_from_checkpoint(path, cfg=Wav2vec2FCModuleConfig(), init_seed=2, **ref_kw, **kw)
        assert b.steps == 123 and b.schedule_step == 123
        assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
        c = Wav2vec2FCModule.load_from_checkpoint(path, cfg=Wav2vec2FCModuleConfig(explicit_num_speakers=11),
                                                  init_seed=3, **ref_kw, **kw)
        sc, sa = c.state_dict(), a.state_dict()
        assert sc["loss_fn.fc_weights"].shape[0] == 11
        assert all(torch.equal(sa[k], sc[k]) for k in sa if k != "loss_fn.fc_weights")
    finally:
        C.W2V2Config.from_huggingface_id = orig



def test_checkpoint_resumes_on_torch_adam_and_one_cycle_lr(tmp_path):
    """the reference wraps its Adam in ``OneCycleLR`` (ref: src/main.py:323-335), which keeps
    ``initial_lr / max_lr / min_lr / base_momentum / max_momentum`` in the optimiser's param group;
    ``Optimizer.load_state_dict`` replaces the groups wholesale, so a checkpoint without them dies in the first
    ``OneCycleLR.step()`` after a resume (KeyError: 'initial_lr').  Here ``optimizer_states[0]`` / ``lr_schedulers[0]`` of a
    saved file go into REAL torch objects, one optimiser + scheduler step runs, and the learning rate / beta1 follow
    this repo's own schedule (which golden g8 pins against torch)."""
    import torch
    from w2v2_speaker_amd import config as C
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import Wav2vec2FCModule, Wav2vec2FCModuleConfig
    from w2v2_speaker_amd.params import ParamStore
    tiny = C.W2V2Config.tiny()
    orig = C.W2V2Config.from_huggingface_id
    C.W2V2Config.from_huggingface_id = staticmethod(lambda _id: tiny)
    try:
        m = Wav2vec2FCModule.from_config(Wav2vec2FCModuleConfig(reset_weights=True), num_speakers=7, init_seed=1,
                                         device="cpu", act_dtype=torch.float32)
        st, sch = m.store, m.schedule
        g = torch.Generator().manual_seed(5)
        st.exp_avg = torch.randn(st.grad.shape, generator=g)
        st.exp_avg_sq = torch.rand(st.grad.shape, generator=g)
        st.step_head = st.step_body = 7
        m.steps = m.schedule_step = 7
        path = str(tmp_path / "resume.ckpt")
        m.save_checkpoint(path)
        ck = torch.load(path, weights_only=True)
        group = ck["optimizer_states"][0]["param_groups"][0]
        assert {"initial_lr", "max_lr", "min_lr", "base_momentum", "max_momentum"} <= set(group)
        assert group["lr"] == sch.at(7)[0] and group["betas"][0] == sch.at(7)[1]          # values of the NEXT step
        params = [torch.nn.Parameter(torch.zeros(tuple(ck["state_dict"][ParamStore.legacy_key(k)].shape)))
                  for k in st.reference_parameter_order()]
        opt = torch.optim.Adam(params, lr=1.0)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=sch.max_lr, total_steps=sch.total_steps)
        opt.load_state_dict(ck["optimizer_states"][0])            # what a PL resume does, in this order
        sched.load_state_dict(ck["lr_schedulers"][0])
        assert sched.last_epoch == 7 and opt.param_groups[0]["lr"] == sch.at(7)[0]
        for q in params:
            q.grad = torch.zeros_like(q)
        opt.step()
        sched.step()                                              # raised KeyError('initial_lr') before round 4
        lr8, b8 = sch.at(8)
        assert abs(opt.param_groups[0]["lr"] - lr8) <= 1e-12 * lr8 + 1e-18
        assert abs(opt.param_groups[0]["betas"][0] - b8) <= 1e-12
        assert sched.get_last_lr()[0] == opt.param_groups[0]["lr"]
    finally:
        C.W2V2Config.from_huggingface_id = orig


def test_checkpoint_keys_and_parameter_order_match_the_reference_stack(tmp_path):
    """The reference's module is ``loss_fn`` + HF ``Wav2Vec2Model`` (under ``wav2vec.model.``) + ``fc_list``
    (ref: wav2vec2_fc.py:101-228).  Key SET of a saved checkpoint == {HF state_dict keys with the torch-1.9 weight-norm
    names} + the head, and ParamStore.reference_parameter_order() == HF ``named_parameters()`` order: the indices of
    the torch-Adam state in ``optimizer_states`` then address the same tensors in the reference's optimiser.  An Adam
    state survives save -> load (moments land at the right arena offsets)."""
    import torch
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    from w2v2_speaker_amd import config as C
    from w2v2_speaker_amd.lightning_modules.speaker.wav2vec2_fc import Wav2vec2FCModule, Wav2vec2FCModuleConfig
    from w2v2_speaker_amd.params import ParamStore, W2V_PREFIX
    tiny = C.W2V2Config.tiny()
    hf = Wav2Vec2Model(Wav2Vec2Config(
        conv_dim=list(tiny.conv_dim), conv_kernel=list(tiny.conv_kernel), conv_stride=list(tiny.conv_stride),
        hidden_size=tiny.hidden_size, num_hidden_layers=tiny.num_hidden_layers,
        num_attention_heads=tiny.num_attention_heads, intermediate_size=tiny.intermediate_size,
        num_conv_pos_embeddings=tiny.num_conv_pos_embeddings,
        num_conv_pos_embedding_groups=tiny.num_conv_pos_embedding_groups))
    hf_order = [W2V_PREFIX + n for n, _ in hf.named_parameters()]
    hf_keys = {ParamStore.legacy_key(W2V_PREFIX + k) for k in hf.state_dict()}
    orig = C.W2V2Config.from_huggingface_id
    C.W2V2Config.from_huggingface_id = staticmethod(lambda _id: tiny)
    try:
        kw = dict(device="cpu", act_dtype=torch.float32)
        m = Wav2vec2FCModule.from_config(Wav2vec2FCModuleConfig(reset_weights=True, hidden_fc_layers_out=[24]),
                                         num_speakers=7, loss="ce", init_seed=1, **kw)
        order = m.store.reference_parameter_order()
        assert [n for n in order if n.startswith(W2V_PREFIX)] == hf_order
        # nn.Module surface: one Parameter per reference parameter, same order and keys as state_dict(), views of the arena
        assert isinstance(m, torch.nn.Module) and [n for n, _ in m.named_parameters()] == order
        assert set(dict(m.named_parameters())) == set(m.state_dict())
        w = dict(m.named_parameters())["fc_list.1.0.bias"]
        w.data.fill_(0.25)
        assert float(m.store.p("fc_list.1.0.bias")[0]) == 0.25 and w.grad.data_ptr() == m.store.g("fc_list.1.0.bias").data_ptr()
        assert m.eval().training is False and m.train().training is True and m.half() is m and m.to("cpu") is m
        assert order[-4:] == ["fc_list.0.0.weight", "fc_list.0.0.bias", "fc_list.1.0.weight", "fc_list.1.0.bias"]
        st = m.store
        g = torch.Generator().manual_seed(3)
        st.exp_avg = torch.randn(st.grad.shape, generator=g)
        st.exp_avg_sq = torch.rand(st.grad.shape, generator=g)
        st.step_head, st.step_body = 9, 5
        path = str(tmp_path / "m.ckpt")
        m.save_checkpoint(path)
        ck = torch.load(path, weights_only=True)         # tensors / ints / containers only: loads without unpickling code
        assert set(ck["state_dict"]) == hf_keys | {"fc_list.0.0.weight", "fc_list.0.0.bias", "fc_list.1.0.weight",
                                                   "fc_list.1.0.bias"}
        osd = ck["optimizer_states"][0]
        for i, n in enumerate(order):                     # state i belongs to reference parameter i
            if i in osd["state"]:
                assert tuple(osd["state"][i]["exp_avg"].shape) == tuple(st.shapes[n])
                assert osd["state"][i]["step"] == (9 if n.startswith("fc_list.") else 5)
        frozen = [i for i, n in enumerate(order) if "feature_extractor" in n]
        assert frozen and not any(i in osd["state"] for i in frozen)       # no gradient ever -> no Adam state (torch)
        ctor_kw = dict(cfg=Wav2vec2FCModuleConfig(hidden_fc_layers_out=[24]), num_speakers=7,
                       loss_fn_constructor=lambda: __import__("w2v2_speaker_amd.optim.loss", fromlist=["x"]).CrossEntropyLoss())
        b = Wav2vec2FCModule.load_from_checkpoint(path, init_seed=2, **ctor_kw, **kw)
        import math
        for name, off in st.offsets.items():              # (the arena pads every tensor to 64 elements)
            if off < st.n_train:
                sl = slice(off, off + math.prod(st.shapes[name]))
                assert torch.equal(b.store.exp_avg[sl], st.exp_avg[sl]), name
                assert torch.equal(b.store.exp_avg_sq[sl], st.exp_avg_sq[sl]), name
        assert (b.store.step_head, b.store.step_body) == (9, 5)
        # a file written with the torch >= 2.1 names loads too
        m.save_checkpoint(path, legacy_weight_norm_names=False)
        c = Wav2vec2FCModule.load_from_checkpoint(path, init_seed=4, **ctor_kw, **kw)
        assert all(torch.equal(v, c.state_dict()[k]) for k, v in m.state_dict().items())
    finally:
        C.W2V2Config.from_huggingface_id = orig


def test_hot_kernels_use_no_scratch(tmp_path):
    """A kernel that touches scratch (private segment) pays extra per dispatch on this stack (tools/probes/
    scratch_probe.hip: 2.8 -> 10.5+ us isolated).  Guard: in the built library only the exact-f32 parity instantiations,
    the f32-output ring GEMM (not launched by the training steps) and the A/B-only 256x256x32 kernel may spill."""
    import re
    import shutil
    import subprocess
    from w2v2_speaker_amd import _build
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(_build.LIB)):
        pytest.skip("needs the ROCm LLVM tools and a built libw2v2hip.so")
    lib = tmp_path / "lib.so"
    shutil.copy(_build.LIB, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, capture_output=True, check=True)
    allowed = re.compile(r"gemm16_ring_256x256_kernel|gemm16_ring_256x128_kernelI\w+fLb0EEv|ln_bwd_kernelIfE|gemm_f32_kernel")
    spills, seen = [], 0
    for f in tmp_path.glob("lib.so.*gfx950"):
        notes = subprocess.run([readelf, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                seen += 1
                if int(m.group(1)) > 0 and not allowed.search(name):
                    spills.append((name, int(m.group(1))))
    assert seen > 100, "kernel metadata not found"
    assert not spills, spills


def test_committed_bench_lines_follow_the_contract():
    """The JSON lines bench.py printed on the GPU box (committed under profiles/, every round's) carry every field the
    driver's contract names; roofline.frac is achieved / peak; the round-3 line says which PMC file its traffic / busy
    figures come from, whether the kernels have changed since, what the matrix pipe sustains (measured) and the bf16
    mode under `also`; the ECAPA line (configs[4]) is at the reference's f32 precision with an HBM roofline per family."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tag in ("r02", "r03"):
        line = json.load(open(os.path.join(root, "profiles", f"{tag}_bench_line.json")))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, (tag, k)
        assert line["unit"] == "utterances/sec" and line["higher_is_better"] is True and line["vs_baseline"] is None
        assert line["dtype"] == "f16" and line["data"] == "synthetic" and "workload" in line["config"]
        assert "model" not in line["config"]
        r = line["roofline"]
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert isinstance(r["traffic"], int) and 0 < r["mfma_busy"] < 1
        assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
        c = line["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
        e = json.load(open(os.path.join(root, "profiles", f"{tag}_ecapa_bench_line.json")))
        assert e["roofline"]["bound"] == "hbm" and e["roofline"]["unit"] == "GB/s" and e["roofline"]["peak"] == 8000.0
        assert e["roofline"]["traffic"] > 0 and len(e["roofline_families"]) >= 3
    # round 3: self-evidencing fields
    assert r["pmc_source"] == "profiles/r03_pmc_counters.json" and r["pmc_stale"] is False
    assert 1500.0 < r["peak_sustained_measured"] < 2500.0
    assert abs(r["frac_of_sustained"] - r["achieved"] / r["peak_sustained_measured"]) < 1e-3
    b16 = line["also"]["bf16"]
    assert b16["unit"] == "utterances/sec" and abs(b16["value"] - 66 / (b16["ms_per_step"] * 1e-3)) < 0.01 * b16["value"]
    assert e["dtype"] == "f32" and e["gemm_mfma"]["peak"] == 157.3 and any(k.startswith("gemm_f32_") for k in e["gemm_mfma"]["by_kernel"])
    pmc = json.load(open(os.path.join(root, "profiles", "r03_pmc_counters.json")))
    assert len(pmc["source_hash"]) == 16 and "gemm16_ring_256x128_kernel<_Float16, _Float16>" in pmc["kernels"]


def test_round6_bench_line_carries_the_tail_families_and_the_centred_eer():
    """profiles/r06_bench_line.json (tools/round_final.sh r06 on an MI355X): besides the contract fields, the non-GEMM tail
    against the HBM roofline (`roofline_families`) and the EER leg's centred figure next to the plain one
    (item 1), both equal to the reference's on the committed line."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", "r06_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "f16" and line["vs_baseline"] is None
    assert abs(line["value"] - 66 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
    r = line["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["pmc_stale"] is False
    assert r["pmc_source"] == "profiles/r06_pmc_counters.json"
    fam = {e["family"]: e for e in line["roofline_families"]}
    assert {"layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd", "adam", "conv0"} <= set(fam)
    for e in fam.values():
        assert e["bound"] == "hbm" and e["peak"] == 8000.0 and e["unit"] == "GB/s"
        assert abs(e["frac"] - e["achieved"] / 8000.0) < 1e-3 and e["avg_us"] > 0 and e["bracket_us"] >= e["avg_us"]
    a = fam["attention_fwd"]
    assert a["rows_padded_to"] == 160 and abs(a["valid_score_fraction"] - 149 * 149 / 160.0 ** 2) < 1e-3 and a["matrix_tflops"] > 100
    assert 10.0 < fam["layernorm_fwd"]["launches_per_step"] < 30 and fam["adam"]["launches_per_step"] == 1.0
    e = line["eer"]
    assert e["hip_f16"] == e["reference"] and e["centred"]["hip_f16"] == e["centred"]["reference"]
    assert e["centred"]["reference"] < 0.5 * e["reference"] and e["centred"]["max_abs_score_diff"] < 5e-3
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0


def test_committed_ddp_rehearsal_line_carries_the_multi_gpu_fields():
    """profiles/r06_ddp_rehearsal_line.json = `W2V2_DIST_BACKEND=gloo W2V2_SHARE_GPU=1 bench.py --gpus 2` on a one-GPU box
    (VERDICT r5 item 8): NOT a scaling measurement (two ranks share one GPU and reduce over gloo) -- it pins the shape of
    the line the driver will read from an N > 1 run: whole-job value, per-rank records, the self-check of the collective."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", "r06_ddp_rehearsal_line.json")))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 2 * 66
    assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
    assert abs(line["utt_per_sec_per_gpu"] * 2 - line["value"]) < 0.01 * line["value"]
    ranks = line["rccl"]["ranks"]
    assert line["rccl"]["backend"] == "gloo" and [r["rank"] for r in ranks] == [0, 1] and all(r["allreduce_ok"] for r in ranks)
    d = line["ddp"]
    for k in ("per_rank_ms_per_step", "per_rank_step_ms_median", "per_rank_solo_ms_per_step"):
        assert len(d[k]) == 2 and all(v > 0 for v in d[k]), k
    assert d["rank_spread_ms"] >= 0 and 0 < d["step_time_ratio_vs_solo"] < 1.5
    assert abs(max(d["per_rank_ms_per_step"]) - line["ms_per_step"]) < 1e-6      # MAX over ranks is what the line reports


def test_f32_weight_gradient_split_factor():
    """ecapa.f32_dw_split: few output tiles -> fill the 512 workgroup slots, at most 32 ways; more tiles than half the
    slots -> the factor with the smallest quantisation loss; never fewer than 256 tokens per split."""
    from w2v2_speaker_amd.ecapa import f32_dw_split
    assert f32_dw_split(128, 384, 19800) == 32             # 3 tiles: capped (atomic contention beyond that)
    assert f32_dw_split(1024, 1024, 19800) == 8            # 64 tiles x 8 = one full round
    assert f32_dw_split(3072, 3072, 19800) == 8            # 576 tiles: 9 full rounds of an eighth instead of 1.125 -> 2
    assert f32_dw_split(128, 3072, 19800) == 21
    assert f32_dw_split(128, 384, 600) == 2                # 256 tokens per split at least
    for n_out, n_in in ((128, 128), (1536, 3072), (4096, 4096), (200, 1024)):
        sk = f32_dw_split(n_out, n_in, 19800)
        assert 1 <= sk <= 32 and 19800 // sk >= 256


def test_prof_summary_steady_state_window(tmp_path):
    """tools/prof_summary.py --steady: only the launches between the (steps+1)-th last and the last launch of the named
    once-per-step kernel are counted, so start-up work does not dilute the per-step table."""
    import sqlite3
    import subprocess
    db = tmp_path / "t.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    rows = [("init_copy", 0, 10)] * 50                                         # start-up
    for step in range(5):
        for k in range(3):
            rows.append(("gemm_kernel(args)", 1000 * (step + 1) + 10 * k, 1000 * (step + 1) + 10 * k + 8))
        rows.append(("adam_kernel<float>(x)", 1000 * (step + 1) + 900, 1000 * (step + 1) + 950))
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), str(db), "3", "--steady", "adam_kernel"],
                         capture_output=True, text=True, check=True).stdout
    assert "steady-state window: the last 3 steps" in out and "init_copy" not in out
    line = [ln for ln in out.splitlines() if "gemm_kernel" in ln][0].split()
    assert line[2] == "9"                                                      # 3 launches x 3 steps
    whole = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), str(db), "5"],
                           capture_output=True, text=True, check=True).stdout
    assert "init_copy" in whole


def test_rank_agreement_signature_is_exact_in_f32_at_base_size():
    """the start-up check of CAbiBucketAllReducer.broadcast_parameters all-reduces an f32 signature.
    With the byte count of wav2vec2-base's replica state (103578496 floats x 4 B, + moments + a loss scale) the former
    13-bit limbs squared past 2^24 and AGREEING ranks failed; 8-bit limbs keep every summed value exact for <= 258 ranks.
    Simulated here exactly as the wire does it: f32 vectors summed over the ranks in f32."""
    from w2v2_speaker_amd.comm import signature_agrees, signature_limbs
    flat = 103578496
    for nbytes in (flat * 4 * 3 + 8, flat * 4 * 3 + flat * 2 + 16, (1 << 48) - 1, 0, 255, 256, 65535 * 8191):
        for world in (2, 4, 8, 64, 258):
            v = torch.tensor(signature_limbs([27, 3, nbytes]), dtype=torch.float32)
            acc = torch.zeros_like(v)
            for _ in range(world):
                acc = acc + v                                   # f32 accumulation, any order gives the same exact sums
            assert float(acc.max()) < 2 ** 24
            assert signature_agrees(acc.tolist(), world), (nbytes, world)
            # ONE rank off by one byte / one tensor / one counter must be caught
            for other in ([27, 3, nbytes ^ 1], [28, 3, nbytes], [27, 2, nbytes], [27, 3, nbytes ^ (1 << 40)]):
                bad = acc - v + torch.tensor(signature_limbs(other), dtype=torch.float32)
                assert not signature_agrees(bad.tolist(), world), (nbytes, world, other)
    with pytest.raises(ValueError):
        signature_limbs([1 << 48])
    with pytest.raises(ValueError):
        signature_agrees([0.0] * 36, 259)


def test_pre_ln_family_parameter_names_order_and_id_table():
    """The "-lv60" / xlsr family (W2V2Config.do_stable_layer_norm / feat_extract_norm="layer" / conv_bias): the arena holds
    exactly HF's parameters for those config flags, reference_parameter_order() == HF ``named_parameters()`` order; and
    from_huggingface_id classifies the ids it knows and RAISES on a 'large' id it cannot classify (VERDICT r5 missing 6)."""
    import dataclasses
    import torch
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    from w2v2_speaker_amd.config import W2V2Config
    from w2v2_speaker_amd.params import ParamStore, W2V_PREFIX
    tiny = dataclasses.replace(W2V2Config.tiny(), do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True)
    hf = Wav2Vec2Model(Wav2Vec2Config(
        conv_dim=list(tiny.conv_dim), conv_kernel=list(tiny.conv_kernel), conv_stride=list(tiny.conv_stride),
        hidden_size=tiny.hidden_size, num_hidden_layers=tiny.num_hidden_layers,
        num_attention_heads=tiny.num_attention_heads, intermediate_size=tiny.intermediate_size,
        num_conv_pos_embeddings=tiny.num_conv_pos_embeddings,
        num_conv_pos_embedding_groups=tiny.num_conv_pos_embedding_groups,
        do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True))
    st = ParamStore(tiny, "cpu", torch.float32, head=None, num_speakers=1)
    hf_named = [(W2V_PREFIX + n, tuple(p.shape)) for n, p in hf.named_parameters()]
    assert [n for n in st.reference_parameter_order()] == [n for n, _ in hf_named]
    assert all(tuple(st.shapes[n]) == s for n, s in hf_named)
    # the feature extractor of this family is frozen storage like the group-norm one: nothing of it is trainable
    assert not any(st.is_trainable(n) for n in st.shapes if "feature_extractor" in n)
    f = W2V2Config.from_huggingface_id
    for hid in ("facebook/wav2vec2-large-lv60", "facebook/wav2vec2-large-960h-lv60-self", "facebook/wav2vec2-large-xlsr-53",
                "facebook/wav2vec2-large-robust", "facebook/wav2vec2-large-100k-voxpopuli"):
        c = f(hid)
        assert (c.hidden_size, c.num_hidden_layers, c.do_stable_layer_norm, c.feat_extract_norm, c.conv_bias) == \
            (1024, 24, True, "layer", True), hid
    for hid in ("facebook/wav2vec2-large", "facebook/wav2vec2-large-960h"):
        c = f(hid)
        assert c.hidden_size == 1024 and not c.do_stable_layer_norm and c.feat_extract_norm == "group" and not c.conv_bias
    assert f("facebook/wav2vec2-base-960h").hidden_size == 768
    for hid in ("someone/wav2vec2-large-finetuned-xyz", "facebook/hubert-xl"):
        with pytest.raises(ValueError):
            f(hid)


def test_no_compiler_vmcnt_inside_the_k_loops_of_the_lds_dma_gemms():
    """Round 6: the compiler's waitcnt pass had its own `s_waitcnt vmcnt(0)` at the top of the steady-state K loop of the ring and
    the phased GEMM (behind their counted waits written in assembly) -- every K tile waited for every LDS-DMA piece in flight.
    tools/loop_waits.sh lists the waits that are NOT inside an inline-assembly block with the loop they sit in; the K loops
    (depth 2, inside the persistent tile loop) must have none.  Cross-compiles two files (about a minute), no GPU."""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src, pat in (("gemm_ring.hip", "IDF16_DF16_Lb0E"), ("gemm_phased.hip", "IDF16_DF16_Li0E")):
        out = subprocess.run(["bash", os.path.join(root, "tools", "loop_waits.sh"), src, pat], capture_output=True, text=True,
                             timeout=600).stdout
        assert "_Z" in out, (src, out[-400:])                   # the kernel was found and compiled
        inner = [l for l in out.splitlines() if "Depth=2" in l or "Depth=3" in l]
        assert not inner, (src, inner[:4])
