"""GPU probe: parity numbers of the three activation dtypes against the reference goldens (prints, no asserts)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, rel_l2  # noqa: E402
from oracle import w2v2_oracle as O  # noqa: E402
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig  # noqa: E402
from w2v2_speaker_amd.engine import Plan  # noqa: E402
from w2v2_speaker_amd.params import ParamStore  # noqa: E402

DEV = "cuda"
T = lambda a: torch.from_numpy(np.asarray(a))
NOREG = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                                     hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)


def store(cfg, ocfg, dtype, C):
    st = ParamStore(cfg, DEV, dtype, head="aam", num_speakers=C)
    sd = O.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (C, st.embed_dim), 20211)
    st.load_state_dict(sd)
    return st


def gs(st):
    return float(st.scaler[0]) if st.scaler is not None else 1.0


def tiny(dtype):
    g = np.load(os.path.join(GOLDEN, "g1_tiny.npz"))
    cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
    st = store(cfg, ocfg, dtype, 10)
    wav, label, mask = T(g["wav"]).to(DEV), T(g["label"]).to(DEV), T(g["mask"])
    plan = Plan(st, 2, wav.shape[-1], train=True, reg=NOREG)
    st.zero_grad()
    emb = plan.embed(wav, mask.to(DEV))
    loss, sm = plan.head_forward_backward(label)
    plan.backward()
    torch.cuda.synchronize()
    B, Tn, H = plan.out.shape
    print(f"tiny {dtype}: conv_out {rel_l2(plan.conv[-1].float().cpu(), g['stage.conv_out']):.2e} "
          f"enc_in {rel_l2(plan.X[0].float().cpu().view(B, Tn, H), g['stage.enc_in']):.2e} "
          f"last {rel_l2(plan.X[-1].float().cpu().view(B, Tn, H), g['stage.layer1']):.2e} emb {rel_l2(emb.cpu(), g['embedding']):.2e} "
          f"loss {float(loss):.5f} vs {float(g['loss']):.5f} softmax {rel_l2(sm.cpu(), g['softmax']):.2e}")
    worst, wn = 0.0, ""
    s = gs(st)
    for name in st.shapes:
        if not st.is_trainable(name):
            continue
        key = "grad." + (name[len("wav2vec.model."):] if name.startswith("wav2vec.model.") else name)
        ref = g[key]
        got = st.g(name).cpu().numpy().astype(np.float64) / s
        e = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-12)
        if e > worst and np.linalg.norm(ref) > 1e-3 * float(np.linalg.norm(g["grad.loss_fn.fc_weights"])):
            worst, wn = e, name
    print(f"   worst grad rel err {worst:.3e} ({wn})  scaler={None if st.scaler is None else st.scaler.tolist()}")


def base(dtype, scale=None):
    g = np.load(os.path.join(GOLDEN, "g2_base.npz"))
    cfg, ocfg = W2V2Config(), O.OracleConfig.base()
    st = store(cfg, ocfg, dtype, 5994)
    if scale is not None and st.scaler is not None:
        st.scaler[0] = scale
    wav, label = O.synth_batch(2, 48000, 5994, seed=42133724)
    wav, label = wav.to(DEV), label.to(DEV)
    ev = Plan(st, 2, 48000, train=False)
    e = ev.embed(wav).clone()
    torch.cuda.synchronize()
    print(f"base {dtype}: eval emb rel-L2 {rel_l2(e.cpu(), g['eval.mean+std']):.3e} (fused={ev.fused}) "
          f"hidden sample {rel_l2(ev.out.float().cpu()[:, ::16, ::16], g['eval.last_hidden.sample']):.3e}")
    cl = Plan(st, 2, 48000, train=False, pooling="first+cls", insert_cls_token=True)
    print(f"   first+cls emb {rel_l2(cl.embed(wav).cpu(), g['eval.first+cls']):.3e}")
    del ev, cl
    tr = Plan(st, 2, 48000, train=True, reg=NOREG)
    st.zero_grad()
    emb = tr.embed(wav, T(g["mask"]).to(DEV))
    loss, sm = tr.head_forward_backward(label)
    tr.backward()
    torch.cuda.synchronize()
    print(f"   train emb {rel_l2(emb.cpu(), g['train.embedding']):.3e} loss {float(loss):.5f} vs {float(g['train.loss']):.5f}")
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    s = gs(st)
    devs = []
    for n, ref in norms.items():
        name = n if n.startswith("loss_fn") else "wav2vec.model." + n
        if not st.is_trainable(name):
            continue
        got = float(st.g(name).double().norm()) / s
        devs.append((abs(got - ref) / (ref + 1e-3 * max(norms.values())), n))
        head = st.g(name).flatten()[:32].cpu().numpy() / s
    devs.sort(reverse=True)
    print("   worst grad-norm deviations:", [(round(a, 4), b) for a, b in devs[:4]])
    n = "loss_fn.fc_weights"
    print("   finite:", bool(torch.isfinite(st.grad).all()), " scaler", None if st.scaler is None else st.scaler.tolist(),
          " max|G|/scale", float(tr.G.float().abs().max()) / s, " max|DQKV|/scale",
          float(tr._gsets[0]["DQKV"].float().abs().max()) / s, float(tr._gsets[0]["DH"].float().abs().max()) / s)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f32", "bf16", "f16"]
    m = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    for w in which:
        tiny(m[w])
    for w in which:
        base(m[w])
    if "f16" in which:
        for sc in (64.0, 1024.0, 4096.0):
            base(torch.float16, sc)
