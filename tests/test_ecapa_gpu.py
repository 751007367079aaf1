"""ECAPA-TDNN path (BASELINE configs[4]) on the HIP kernels against oracle/ecapa_oracle.py.
The oracle restates speechbrain's published ECAPA_TDNN (speechbrain is not available here): parity unpinned."""
import math

import pytest
import torch

from conftest import rel_l2
from oracle import ecapa_oracle as E
from oracle import w2v2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(dtype, B=4, T=50, classes=9):
    from w2v2_speaker_amd.ecapa import EcapaConfig, EcapaPlan, EcapaStore
    cfg, ocfg = EcapaConfig.tiny(), E.EcapaConfig.tiny()
    st = EcapaStore(cfg, DEV, dtype, num_speakers=classes)
    sd = E.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (classes, cfg.lin_neurons), 20211)
    st.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(B, T, cfg.input_mel_coefficients, generator=g)
    label = torch.randint(0, classes, (B,), generator=g)
    return cfg, ocfg, st, sd, feat, label


def test_reflect_im2col_and_its_adjoint():
    from w2v2_speaker_amd import ops as o
    B, T, C, k, d = 2, 23, 16, 5, 3
    x = torch.randn(B, T, C)
    col = torch.zeros(B * T, k * C, device=DEV)
    o.im2col_reflect(x.to(DEV).view(B * T, C), C, col, B, T, C, k, d)
    p = d * (k - 1) // 2
    xp = torch.nn.functional.pad(x.transpose(1, 2), (p, p), mode="reflect").transpose(1, 2)      # [B, T+2p, C]
    ref = torch.stack([xp[:, j * d:j * d + T] for j in range(k)], dim=2).reshape(B * T, k * C)
    torch.cuda.synchronize()
    assert torch.equal(col.cpu(), ref)
    # adjoint: <im2col(x), g> == <x, col2im(g)>
    g = torch.randn(B * T, k * C)
    dx = torch.zeros(B * T, C, device=DEV)
    o.col2im_reflect(g.to(DEV), dx, C, B, T, C, k, d, False)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    xpr = torch.nn.functional.pad(xr.transpose(1, 2), (p, p), mode="reflect").transpose(1, 2)
    (torch.stack([xpr[:, j * d:j * d + T] for j in range(k)], dim=2).reshape(B * T, k * C) * g).sum().backward()
    assert rel_l2(dx.cpu().view(B, T, C), xr.grad) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ecapa_training_step_vs_unpinned_restatement(dtype):
    from w2v2_speaker_amd.ecapa import FE, EcapaPlan
    cfg, ocfg, st, sd, feat, label = _setup(dtype)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    emb_ref, stages = E.ecapa_forward(feat, sdg, ocfg, return_stages=True)
    loss_ref, _ = O.aam_softmax(emb_ref, sdg["loss_fn.fc_weights"], label)
    loss_ref.backward()
    plan = EcapaPlan(st, feat.shape[0], feat.shape[1], train=True)
    st.zero_grad()
    emb = plan.embed(feat.to(DEV))
    loss, _ = plan.head_forward_backward(label.to(DEV))
    plan.backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    B, T = feat.shape[:2]
    assert rel_l2(plan.x0.float().cpu().view(B, T, -1), stages["block0"].detach()) < (2e-5 if f32 else 2e-2)
    C1 = cfg.channels[1]
    for i in (1, 2, 3):
        got = plan.cat[:, (i - 1) * C1:i * C1].float().cpu().view(B, T, C1)
        assert rel_l2(got, stages[f"block{i}"].detach()) < (5e-5 if f32 else 3e-2), i
    assert rel_l2(plan.mfa_out.float().cpu().view(B, T, -1), stages["mfa"].detach()) < (5e-5 if f32 else 3e-2)
    assert rel_l2(plan.pooled.cpu(), stages["asp"].detach()) < (5e-5 if f32 else 3e-2)
    assert rel_l2(emb.cpu(), emb_ref.detach()) < (1e-4 if f32 else 6e-2)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < (1e-4 if f32 else 5e-2) * abs(float(loss_ref.detach()))
    if not f32:
        # bf16: the chained BatchNorm backward amplifies rounding noise, so end-to-end gradients are checked in f32;
        # here every weight / bias gradient of the bf16-only GROUPED wgrad path is checked against an f64 product of
        # the plan's own stored operands (da and the conv input / im2col buffer)
        for t in plan._tdnns():
            da = t.da.double().cpu()
            A = (t.col if t.k > 1 else t.x[:, :t.cin]).double().cpu()
            dw_ref = (da.t() @ A).view(t.cout, t.k, t.cin).permute(0, 2, 1)
            assert rel_l2(st.g(t.pre + "conv.conv.weight").cpu(), dw_ref) < 1e-4, t.pre
            got_b, ref_b = st.g(t.pre + "conv.conv.bias").double().cpu(), da.sum(0)
            assert float((got_b - ref_b).abs().max()) < 1e-4 * float(da.abs().sum(0).max()) + 1e-7, t.pre
        return
    gmax = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    bad = []
    for n, v in sdg.items():
        name = n if n.startswith("loss_fn") else FE + n
        got, ref = st.g(name).double().cpu().reshape(v.grad.shape), v.grad.double()
        err = float((got - ref).norm())
        if err > 3e-3 * float(ref.norm()) + 2e-6 * gmax:
            bad.append((n, round(err, 6), round(float(ref.norm()), 6)))
    assert not bad, bad[:8]


def test_ecapa_batchnorm_running_statistics_are_shared_saved_and_used_by_eval():
    """BatchNorm running statistics belong to the store, not to a plan: the evaluation plan must
    normalise with what the training plan accumulated, and checkpoints must carry them under the speechbrain names."""
    from w2v2_speaker_amd.ecapa import FE, EcapaPlan, EcapaStore
    cfg, ocfg, st, sd, feat, label = _setup(torch.float32, B=4, T=50, classes=4)
    tr = EcapaPlan(st, 4, 50, train=True)
    ev = EcapaPlan(st, 4, 50, train=False)
    for a, b in zip(tr._tdnns(), ev._tdnns()):
        assert a.running.data_ptr() == b.running.data_ptr() == st.running(a.pre + "norm.norm.weight").data_ptr()
    assert tr.bn_running.data_ptr() == ev.bn_running.data_ptr()
    e_before = ev.embed(feat.to(DEV)).clone()
    tr.embed(feat.to(DEV))                                   # one training forward: momentum 0.1 update
    torch.cuda.synchronize()
    t0 = tr._tdnns()[0]
    act = torch.relu(t0.a.float()).double().cpu()            # BatchNorm input of block 0 (ReLU of the saved conv output)
    C = t0.cout
    assert rel_l2(t0.running[:C].cpu(), 0.1 * act.mean(0)) < 1e-4
    assert rel_l2(t0.running[C:].cpu(), 0.9 + 0.1 * act.var(0, unbiased=True)) < 1e-4
    e_after = ev.embed(feat.to(DEV)).clone()
    torch.cuda.synchronize()
    assert rel_l2(e_after.cpu(), e_before.cpu()) > 1e-3      # the evaluation plan sees the updated statistics
    full = st.state_dict()
    key = FE + "blocks.0.norm.norm.running_mean"
    assert key in full and FE + "asp_bn.norm.running_var" in full and rel_l2(full[key], t0.running[:C].cpu()) == 0.0
    st2 = EcapaStore(cfg, DEV, torch.float32, num_speakers=4)
    st2.load_state_dict(full)
    ev2 = EcapaPlan(st2, 4, 50, train=False)
    assert rel_l2(ev2.embed(feat.to(DEV)).cpu(), e_after.cpu()) < 1e-6


def test_ecapa_trainer_reduces_loss():
    from w2v2_speaker_amd.ecapa import EcapaPlan, EcapaTrainer
    from w2v2_speaker_amd.optim.schedule import Constant
    cfg, ocfg, st, sd, feat, label = _setup(torch.bfloat16, B=8, T=60, classes=4)
    plan = EcapaPlan(st, 8, 60, train=True)
    tr = EcapaTrainer(st, plan, Constant(2e-3, 0.9))
    losses = [float(tr.train_step(feat.to(DEV), label.to(DEV))[0]) for _ in range(25)]
    assert all(math.isfinite(l) for l in losses) and losses[-1] < 0.6 * losses[0], losses


@pytest.mark.parametrize("B,N,K", [(66, 128, 1024), (66, 1024, 128), (66, 192, 6144), (5, 24, 20), (1, 3, 4), (33, 130, 260)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_skinny_linear_kernels_vs_torch(B, N, K, act):
    """csrc/skinny.hip (SE bottleneck, ECAPA fc, wav2vec2 fc_list): forward and both gradients against torch in f64."""
    from w2v2_speaker_amd import ops
    g = torch.Generator().manual_seed(B * 131 + N + K + act)
    x, W, b = torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    dy = torch.randn(B, N, generator=g)
    xr, Wr, br = x.double().requires_grad_(), W.double().requires_grad_(), b.double().requires_grad_()
    pre = xr @ Wr.t() + br
    yr = pre if act == 0 else torch.relu(pre) if act == 1 else torch.sigmoid(pre)
    yr.backward(dy.double())
    dev = "cuda"
    xd, Wd, bd, dyd = x.to(dev), W.to(dev), b.to(dev), dy.to(dev)
    y = torch.empty(B, N, device=dev)
    ops.skinny_linear_fwd(xd, Wd, bd, y, act)
    assert rel_l2(y.cpu().double(), yr.detach()) < 2e-6
    dx = torch.empty(B, K, device=dev)
    ops.skinny_linear_bwd_x(dyd, y if act else None, Wd, dx, act)
    assert rel_l2(dx.cpu().double(), xr.grad) < 5e-6
    dW, db = torch.full((N, K), 0.5, device=dev), torch.full((N,), -0.25, device=dev)
    ops.skinny_linear_bwd_w(dyd, y if act else None, xd, dW, db, act, True)         # accumulate onto the fill values
    assert rel_l2(dW.cpu().double() - 0.5, Wr.grad) < 5e-6
    assert rel_l2(db.cpu().double() + 0.25, br.grad) < 5e-6
    ops.skinny_linear_bwd_w(dyd, y if act else None, xd, dW, db, act, False)        # overwrite
    assert rel_l2(dW.cpu().double(), Wr.grad) < 5e-6 and rel_l2(db.cpu().double(), br.grad) < 5e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,ld,relu", [(9900, 128, 1024, True), (3300, 1024, 1024, True), (66, 6144, 6144, False),
                                          (257, 8, 8, True), (130, 200, 208, False),
                                          (2050, 3072, 3072, False), (1501, 1024, 1280, True), (1100, 1536, 1536, True)])
def test_batchnorm_kernels_vs_torch(dtype, M, C, ld, relu):
    """csrc/tdnn.hip BatchNorm (partial sums + fold/apply) on a row-strided [M, C] view against torch BatchNorm1d
    in f64: output, running statistics, input gradient, dgamma / dbeta; then the evaluation mode.  C >= 1024 with
    M >= 1024 runs the whole-row apply kernels behind a finalize launch (round 6), the rest the 128-channel strip kernels."""
    from w2v2_speaker_amd import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(M + C)
    a = (torch.randn(M, ld, generator=g) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(M, ld, generator=g).to(dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ar = a[:, :C].double().requires_grad_()
    bn = torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.1).double()
    bn.weight.data.copy_(gamma)
    bn.bias.data.copy_(beta)
    yr = bn(torch.relu(ar) if relu else ar)
    yr.backward(dy[:, :C].double())
    ad, dyd = a.to(dev), dy.to(dev)
    y, da = torch.zeros(M, ld, dtype=dtype, device=dev), torch.zeros(M, ld, dtype=dtype, device=dev)
    work, mr = ops.bn_workspace(M, C, dev), torch.empty(C, 2, device=dev)
    running = torch.cat([torch.zeros(C), torch.ones(C)]).to(dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    ops.bn_fwd(ad, ld, work, mr, running, gd, bd, y, ld, M, C, 1e-5, 0.1, relu, True)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_l2(y[:, :C].cpu().double(), yr.detach()) < tol
    assert float(y[:, C:].abs().max()) == 0.0 if ld > C else True               # columns outside the view untouched
    assert rel_l2(running[:C].cpu().double(), bn.running_mean) < 1e-5
    assert rel_l2(running[C:].cpu().double(), bn.running_var) < 1e-5
    dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
    csp = torch.full((ops.bn_colsum_rows(M, C), C), float("nan"), device=dev)
    ops.bn_bwd(dyd, ld, ad, ld, mr, gd, work, dgam, dbet, da, ld, M, C, relu, colsum_partial=csp)
    gtol = 2e-5 if dtype == torch.float32 else 8e-3
    assert rel_l2(da[:, :C].cpu().double(), ar.grad) < gtol
    # fused column sums of da (the bias gradient of the convolution in front): per row block, of the STORED values
    cs_ref = da[:, :C].float().cpu().double().sum(0)
    ctol = 1e-4 if dtype == torch.float32 else 1e-2        # (16-bit: the sums are taken BEFORE the values are rounded for the store)
    assert float((csp.cpu().double().sum(0) - cs_ref).abs().max()) <= ctol * float(da[:, :C].float().abs().sum(0).max().cpu()) + 1e-6
    assert rel_l2(dgam.cpu().double(), bn.weight.grad) < 1e-4 and rel_l2(dbet.cpu().double(), bn.bias.grad) < 1e-4
    bn.eval()
    ye = bn(torch.relu(ar) if relu else ar)
    run0 = running.clone()
    ops.bn_fwd(ad, ld, None, mr, running, gd, bd, y, ld, M, C, 1e-5, 0.1, relu, False)
    assert rel_l2(y[:, :C].cpu().double(), ye.detach()) < tol
    assert torch.equal(running, run0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_operand_forms_of_im2col_and_batchnorm_backward_equal_the_materialised_sum(dtype):
    """w2v2_im2col_reflect_sum / w2v2_bn_bwd_sum (round 5: the Res2Net chunk input x_i + y_{i-1} and its two-source output
    gradient are summed by the kernel that reads them) against add_strided followed by the one-operand kernel.  f32:
    bit-equal.  bf16: the im2col form rounds the same sum once, so it is bit-equal too; the BatchNorm backward adds in
    f32 what the old path had rounded to bf16 first, so it is compared with the f32 pipeline instead."""
    from w2v2_speaker_amd import ops
    dev = DEV
    B, T, w, C, k, dil = 3, 57, 16, 64, 3, 2
    M = B * T
    g = torch.Generator().manual_seed(5)
    t1 = torch.randn(M, C, generator=g).to(dtype).to(dev)
    r2 = torch.randn(M, C, generator=g).to(dtype).to(dev)
    x, x2 = t1[:, 2 * w:3 * w], r2[:, w:2 * w]
    summed = torch.empty(M, w, dtype=dtype, device=dev)
    ops.add_strided(x, C, x2, C, summed, w, M, w)
    col_ref = torch.empty(M, k * w, dtype=dtype, device=dev)
    col = torch.empty(M, k * w, dtype=dtype, device=dev)
    ops.im2col_reflect(summed, w, col_ref, B, T, w, k, dil)
    ops.im2col_reflect(x, C, col, B, T, w, k, dil, x2, C)
    assert torch.equal(col, col_ref)
    # BatchNorm backward with dy = d1 + d2
    a = (torch.randn(M, w, generator=g) * 1.5).to(dtype).to(dev)
    d1 = torch.randn(M, C, generator=g).to(dtype).to(dev)
    d2 = torch.randn(M, C, generator=g).to(dtype).to(dev)
    gamma, beta = (torch.rand(w, generator=g) + 0.5).to(dev), torch.randn(w, generator=g).to(dev)
    work, mr = ops.bn_workspace(M, w, dev), torch.empty(w, 2, device=dev)
    y = torch.empty(M, w, dtype=dtype, device=dev)
    ops.bn_fwd(a, w, work, mr, None, gamma, beta, y, w, M, w, 1e-5, 0.1, True, True)

    def run(two_operands, dt_):
        da = torch.empty(M, w, dtype=dt_, device=dev)
        dg, db = torch.empty(w, device=dev), torch.empty(w, device=dev)
        csp = torch.empty(ops.bn_colsum_rows(M, w), w, device=dev)
        a_, y1, y2 = a.to(dt_), d1.to(dt_), d2.to(dt_)
        if two_operands:
            ops.bn_bwd(y1[:, w:2 * w], C, a_, w, mr, gamma, work, dg, db, da, w, M, w, True, colsum_partial=csp,
                       dy2=y2[:, 3 * w:4 * w], lddy2=C)
        else:
            s_ = torch.empty(M, w, dtype=dt_, device=dev)
            ops.add_strided(y1[:, w:2 * w], C, y2[:, 3 * w:4 * w], C, s_, w, M, w)
            ops.bn_bwd(s_, w, a_, w, mr, gamma, work, dg, db, da, w, M, w, True, colsum_partial=csp)
        return da, dg, db, csp

    if dtype == torch.float32:          # the same two-operand form on the whole-row kernels (C >= 1024, M >= 1024)
        Mw, Cw = 1200, 1024
        aw = torch.randn(Mw, Cw, generator=g).to(dev)
        e1, e2 = torch.randn(Mw, Cw, generator=g).to(dev), torch.randn(Mw, Cw, generator=g).to(dev)
        gw, bw = (torch.rand(Cw, generator=g) + 0.5).to(dev), torch.randn(Cw, generator=g).to(dev)
        workw, mrw = ops.bn_workspace(Mw, Cw, dev), torch.empty(Cw, 2, device=dev)
        yw = torch.empty(Mw, Cw, device=dev)
        ops.bn_fwd(aw, Cw, workw, mrw, None, gw, bw, yw, Cw, Mw, Cw, 1e-5, 0.1, True, True)
        outs = []
        for two in (True, False):
            daw = torch.empty(Mw, Cw, device=dev)
            dgw, dbw = torch.empty(Cw, device=dev), torch.empty(Cw, device=dev)
            cw = torch.empty(ops.bn_colsum_rows(Mw, Cw), Cw, device=dev)
            if two:
                ops.bn_bwd(e1, Cw, aw, Cw, mrw, gw, workw, dgw, dbw, daw, Cw, Mw, Cw, True, colsum_partial=cw, dy2=e2, lddy2=Cw)
            else:
                ops.bn_bwd(e1 + e2, Cw, aw, Cw, mrw, gw, workw, dgw, dbw, daw, Cw, Mw, Cw, True, colsum_partial=cw)
            outs.append((daw, dgw, dbw, cw))
        for u, v in zip(*outs):
            assert torch.equal(u, v)
    got = run(True, dtype)
    if dtype == torch.float32:
        for u, v in zip(got, run(False, torch.float32)):
            assert torch.equal(u, v)
    else:
        ref = run(False, torch.float32)                        # (a, d1, d2 hold bf16 values exactly)
        assert rel_l2(got[0].float().cpu().double(), ref[0].cpu().double()) < 6e-3
        assert rel_l2(got[1].cpu().double(), ref[1].cpu().double()) < 1e-4 and rel_l2(got[2].cpu().double(), ref[2].cpu().double()) < 1e-4


def test_ecapa_full_size_f32_step_every_weight_gradient_and_running_statistics():
    """BASELINE configs[4] at ITS OWN size under -m gpu: C = 1024, 66 utterances x 300 frames, f32 --
    the geometry whose token-long weight gradients take the split-K path with f32 atomics (3072 x 3072 x 19800 eight
    ways, the 128-channel Res2Net chunks 32 ways on 64 x 64 tiles) that the tiny geometry never reaches.  One training
    step: finite loss / embedding; the weight gradient of EVERY TDNN block against an f64 product of the operands the
    plan itself stored (da^T x conv input), its bias gradient (the column sums the BatchNorm backward leaves) against
    the f64 column sum of da; BatchNorm running statistics against torch's update rule on the stored pre-activations."""
    from w2v2_speaker_amd.ecapa import BN_EPS, BN_MOMENTUM, FE, EcapaConfig, EcapaPlan, EcapaStore
    cfg = EcapaConfig()
    assert cfg.channels[1] == 1024
    st = EcapaStore(cfg, DEV, torch.float32, num_speakers=5994)
    st.init_weights(1)
    B, T = 66, 300
    plan = EcapaPlan(st, B, T, train=True)
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(B, T, cfg.input_mel_coefficients, generator=g).to(DEV)
    label = torch.randint(0, 5994, (B,), generator=g).to(DEV)
    tdnns = plan._tdnns()
    run0 = {t.pre: t.running.clone() for t in tdnns}
    st.zero_grad()
    emb = plan.embed(feat)
    loss, _ = plan.head_forward_backward(label)
    plan.backward()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and 5.0 < float(loss) < 25.0 and bool(torch.isfinite(emb).all())
    assert bool(torch.isfinite(st.grad).all())
    M = B * T
    worst = 0.0
    for t in tdnns:
        A = t.col if t.k > 1 else t.x[:, :t.cin]
        ref = t.da.double().t() @ A.double()                                   # [cout, K] (K = tap-major, cin-minor)
        if t.k > 1:
            got = t.dwp
        else:
            got = st.g(t.pre + "conv.conv.weight").view(t.cout, t.cin)
        err = float((got.double() - ref).norm() / ref.norm())
        worst = max(worst, err)
        assert err < 2e-5, (t.pre, t.cout, t.K, err)
        if t.k > 1:      # and the packed gradient reached the arena in torch's [cout][cin][tap] layout
            arena = st.g(t.pre + "conv.conv.weight").view(t.cout, t.cin, t.k)
            assert float((arena.permute(0, 2, 1).reshape(t.cout, t.K).double() - ref).norm() / ref.norm()) < 2e-5, t.pre
        db_ref = t.da.double().sum(0)
        assert float((st.g(t.pre + "conv.conv.bias").double() - db_ref).norm() / db_ref.norm().clamp_min(1e-30)) < 1e-4, t.pre
        # running statistics: torch BatchNorm1d(momentum) on relu(a): mean, UNBIASED variance
        r = torch.relu(t.a.double())
        mean, var = r.mean(0), r.var(0, unbiased=True)
        C = t.cout
        exp_mean = (1 - BN_MOMENTUM) * run0[t.pre][:C].double() + BN_MOMENTUM * mean
        exp_var = (1 - BN_MOMENTUM) * run0[t.pre][C:].double() + BN_MOMENTUM * var
        assert float((t.running[:C].double() - exp_mean).abs().max()) < 1e-5 * float(exp_mean.abs().max() + 1), t.pre
        assert float((t.running[C:].double() - exp_var).abs().max()) < 1e-5 * float(exp_var.abs().max() + 1), t.pre
    print(f"full-size ECAPA f32: loss {float(loss):.4f}, worst weight-gradient rel-L2 vs f64 {worst:.2e} over {len(tdnns)} blocks")


import os as _os
import numpy as _np
from conftest import GOLDEN as _GOLDEN
_SB_DIR = _os.environ.get("W2V2_SB_GOLDEN_DIR", _GOLDEN)


@pytest.mark.skipif(not _os.path.exists(_os.path.join(_SB_DIR, "g15_sb_ecapa_tiny.npz")),
                    reason="no speechbrain golden (tests/golden/make_sb_goldens.py needs speechbrain): row a19 stays unpinned")
def test_ecapa_vs_speechbrain_golden():
    """SURVEY 8a row a19 against the REAL speechbrain ECAPA_TDNN (g15_sb_ecapa_tiny.npz, the tiny widths; ref:
    src/lightning_modules/speaker/ecapa_tdnn.py:75-85): stages, embedding and every parameter gradient of the HIP path
    (exact-f32 mode) for the golden's upstream gradient on the embedding."""
    from w2v2_speaker_amd.ecapa import FE, EcapaConfig, EcapaPlan, EcapaStore
    g = _np.load(_os.path.join(_SB_DIR, "g15_sb_ecapa_tiny.npz"), allow_pickle=False)
    cfg, ocfg = EcapaConfig.tiny(), E.EcapaConfig.tiny()
    st = EcapaStore(cfg, DEV, torch.float32, num_speakers=9)
    sd = E.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (9, cfg.lin_neurons), 20211)
    st.load_state_dict(sd)
    feat = torch.from_numpy(g["feat"])
    B, T = feat.shape[:2]
    plan = EcapaPlan(st, B, T, train=True)
    st.zero_grad()
    emb = plan.embed(feat.to(DEV))
    plan.head.demb.copy_(torch.from_numpy(g["upstream"]).to(DEV))          # the golden's loss is <embedding, upstream>
    plan.backward()
    torch.cuda.synchronize()
    assert rel_l2(plan.x0.float().cpu().view(B, T, -1), g["stage.block0"]) < 2e-5
    C1 = cfg.channels[1]
    for i in (1, 2, 3):
        assert rel_l2(plan.cat[:, (i - 1) * C1:i * C1].float().cpu().view(B, T, C1), g[f"stage.block{i}"]) < 5e-5, i
    assert rel_l2(plan.mfa_out.float().cpu().view(B, T, -1), g["stage.mfa"]) < 5e-5
    assert rel_l2(plan.pooled.cpu(), g["stage.asp"]) < 5e-5
    assert rel_l2(emb.cpu(), g["embedding"]) < 1e-4
    gmax = max(float(_np.linalg.norm(g[k])) for k in g.files if k.startswith("grad."))
    bad = []
    for k in g.files:
        if not k.startswith("grad."):
            continue
        got = st.g(FE + k[len("grad."):]).double().cpu().numpy().reshape(g[k].shape)
        err = float(_np.linalg.norm(got - g[k]))
        if err > 3e-3 * float(_np.linalg.norm(g[k])) + 2e-6 * gmax:
            bad.append((k, err))
    assert not bad, bad[:8]
