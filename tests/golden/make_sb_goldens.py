#!/usr/bin/env python3
"""Five-minute pin for SURVEY rows a10 (attentive statistics pooling) and a19 (ECAPA-TDNN): generate goldens from the
REAL speechbrain classes, wherever ``import speechbrain`` works (it does not in the authoring container: no network,
package absent -- that is why those two rows are "parity unpinned").

    pip install speechbrain            # on any host that has it: the reference pins ^0.5.5 (pyproject.toml:31)
    python tests/golden/make_sb_goldens.py            # -> tests/golden/g15_sb_{asp,ecapa_tiny,seres2net}.npz
    python -m pytest tests -k speechbrain             # CPU: the oracle restatement against them; -m gpu: the HIP path

What is run (the reference's own call sites):
  * ``AttentiveStatisticsPooling(768)`` exactly as ``AttentiveStatPool1D`` builds it (ref: src/layers/pooling.py:87-106),
    C = 768, T = 149, B = 4, training mode (BatchNorm batch statistics): output [B, 2C] + gradients of the input and of
    every parameter for a fixed upstream gradient;
  * ``ECAPA_TDNN`` with the reference's constructor arguments (ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85,
    config/network/ecapa_tdnn.yaml:4-30) at the tiny width of ``oracle.ecapa_oracle.EcapaConfig.tiny()``: per-stage
    activations (forward hooks), embedding, every parameter gradient;
  * ONE ``SERes2NetBlock`` at full width (1024 channels, scale 8, SE 128, k 3, dilation 2), B = 2, T = 60: output + gradients.
Weights come from the name-keyed PCG64 generator the other goldens use (oracle.ecapa_oracle.make_state_dict /
w2v2_speaker_amd.data.synthetic.synth_weight), so nothing but small arrays is written; the speechbrain version and the file
the classes came from are recorded in each .npz (``sb_version``, ``sb_file``).

``--self-test DIR`` runs the same code with the ORACLE's restatement standing in for the speechbrain classes and writes
to DIR: it checks this script's plumbing (key mapping, hooks, file layout, the consuming tests) where speechbrain is
absent.  Files written that way carry ``sb_version = "self-test"`` and pin nothing -- never commit them.
"""
import argparse
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import ecapa_oracle as E
from oracle import w2v2_oracle as O

ASP_C, ASP_T, ASP_B = 768, 149, 4


def to_np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}






def asp_state(C, A=128, seed=20211):
    """speechbrain state-dict names of AttentiveStatisticsPooling -> synthetic parameters (same scaling rules as
    oracle.ecapa_oracle.make_state_dict)."""
    shp = {"tdnn.conv.conv.weight": (A, 3 * C, 1), "tdnn.conv.conv.bias": (A,), "tdnn.norm.norm.weight": (A,),
           "tdnn.norm.norm.bias": (A,), "conv.conv.weight": (C, A, 1), "conv.conv.bias": (C,)}
    sd = {}
    for n, s in shp.items():
        t = O.synth_tensor("sbasp." + n, s, seed)
        if n.endswith("norm.weight"):
            t = 1.0 + 0.1 * t
        elif n.endswith("bias"):
            t = 0.1 * t
        else:
            t = t * (1.5 / (s[1] * s[2]) ** 0.5)
        sd[n] = t
    return sd


def load_params(module, sd):
    """Copy ``sd`` into a speechbrain module by state-dict name; every PARAMETER of the module must be covered."""
    own = dict(module.named_parameters())
    missing = [n for n in own if n not in sd]
    extra = [n for n in sd if n not in own]
    if missing or extra:
        raise SystemExit(f"state-dict names differ from this speechbrain version: missing {missing[:4]} unexpected {extra[:4]}")
    with torch.no_grad():
        for n, p in own.items():
            assert tuple(p.shape) == tuple(sd[n].shape), (n, tuple(p.shape), tuple(sd[n].shape))
            p.copy_(sd[n])


# --------------------------------------------------------------------------------------------- oracle stand-ins (self-test)
class _OracleASP(torch.nn.Module):
    def __init__(self, C, attention_channels=128, global_context=True):
        super().__init__()
        for n, t in asp_state(C, attention_channels).items():
            self.register_parameter(n.replace(".", "__"), torch.nn.Parameter(torch.zeros_like(t)))

    def named_parameters(self, *a, **k):
        for n, p in super().named_parameters(*a, **k):
            yield n.replace("__", "."), p

    def forward(self, x):
        sd = dict(self.named_parameters())
        asp = {"tdnn.conv.weight": sd["tdnn.conv.conv.weight"], "tdnn.conv.bias": sd["tdnn.conv.conv.bias"],
               "tdnn.norm.weight": sd["tdnn.norm.norm.weight"], "tdnn.norm.bias": sd["tdnn.norm.norm.bias"],
               "conv.weight": sd["conv.conv.weight"], "conv.bias": sd["conv.conv.bias"]}
        return O.attentive_stat_pool(x.transpose(1, 2), asp)[:, :, None]


def get_classes(self_test: bool):
    if self_test:
        return None, "self-test", "oracle/ecapa_oracle.py"
    try:
        import speechbrain
        from speechbrain.lobes.models import ECAPA_TDNN as sbm
    except Exception as ex:
        raise SystemExit(f"speechbrain does not import here ({ex!r}): nothing generated.  Install it, or use --self-test DIR "
                         "to exercise the plumbing only.")
    return sbm, getattr(speechbrain, "__version__", "unknown"), getattr(sbm, "__file__", "?")


# --------------------------------------------------------------------------------------------- the three goldens
def golden_asp(sbm, meta, out):
    """ref: src/layers/pooling.py:87-106 -- AttentiveStatPool1D(dim_to_reduce=2, embedding_size=768): the layer gets
    [B, C, T] and returns [B, 2C, 1] -> squeeze."""
    C, T, B = ASP_C, ASP_T, ASP_B
    pool = (sbm.AttentiveStatisticsPooling(C) if sbm is not None else _OracleASP(C))
    sd = asp_state(C)
    load_params(pool, sd)
    pool.train()
    x = O.synth_tensor("sbasp.x", (B, T, C), 31).requires_grad_(True)          # [B, T, C] as the encoder hands it over
    up = O.synth_tensor("sbasp.up", (B, 2 * C), 32)
    y = pool(x.transpose(1, 2)).squeeze(-1)                                  # pooling.py:101-106
    (y * up).sum().backward()
    g = {"x": x.detach(), "upstream": up, "out": y, "dx": x.grad}
    for n, p in pool.named_parameters():
        g["param." + n], g["grad." + n] = sd[n], p.grad
    np.savez_compressed(os.path.join(out, "g15_sb_asp.npz"), **to_np(g), **meta)
    print("g15_sb_asp: out norm", float(y.norm()))


def golden_ecapa_tiny(sbm, meta, out):
    """ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85 with the tiny widths of EcapaConfig.tiny()."""
    cfg = E.EcapaConfig.tiny()
    sd = E.make_state_dict(cfg, 20211)
    B, T = 4, 50
    feat = O.synth_tensor("sbecapa.feat", (B, T, cfg.input_size), 41).requires_grad_(True)
    up = O.synth_tensor("sbecapa.up", (B, cfg.lin_neurons), 42)
    stages = {}
    if sbm is not None:
        net = sbm.ECAPA_TDNN(input_size=cfg.input_size, lin_neurons=cfg.lin_neurons, channels=list(cfg.channels),
                             kernel_sizes=list(cfg.kernel_sizes), dilations=list(cfg.dilations),
                             attention_channels=cfg.attention_channels, res2net_scale=cfg.res2net_scale,
                             se_channels=cfg.se_channels)
        load_params(net, sd)
        net.train()
        hooks = []
        for i, blk in enumerate(net.blocks):
            hooks.append(blk.register_forward_hook(
                lambda _m, _i, o, i=i: stages.__setitem__(f"block{i}", (o[0] if isinstance(o, tuple) else o).detach().transpose(1, 2))))
        hooks.append(net.mfa.register_forward_hook(lambda _m, _i, o: stages.__setitem__("mfa", o.detach().transpose(1, 2))))
        hooks.append(net.asp.register_forward_hook(lambda _m, _i, o: stages.__setitem__("asp", o.detach().squeeze(-1))))
        emb = net(feat).squeeze(1)                                           # [B, 1, lin] -> [B, lin]
        (emb * up).sum().backward()
        for h in hooks:
            h.remove()
        grads = {n: p.grad for n, p in net.named_parameters()}
    else:
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        emb, st = E.ecapa_forward(feat, sdg, cfg, return_stages=True)
        stages = {k: v.detach() for k, v in st.items()}
        (emb * up).sum().backward()
        grads = {n: v.grad for n, v in sdg.items()}
    g = {"feat": feat.detach(), "upstream": up, "embedding": emb, "dfeat": feat.grad}
    for k, v in stages.items():
        g["stage." + k] = v
    for n, gr in grads.items():
        g["grad." + n] = gr if gr is not None else torch.zeros_like(sd[n])
    np.savez_compressed(os.path.join(out, "g15_sb_ecapa_tiny.npz"), **to_np(g), **meta)
    print("g15_sb_ecapa_tiny: emb norm", float(emb.norm()), "stages", sorted(stages))


def golden_seres2net(sbm, meta, out):
    """One SE-Res2Net block at the reference's full width (config/network/ecapa_tdnn.yaml: 1024 channels, scale 8,
    SE 128, kernel 3, dilation 2)."""
    cfg = E.EcapaConfig()
    C, B, T, dil = cfg.channels[1], 2, 60, cfg.dilations[1]
    full = E.make_state_dict(cfg, 20211)
    sd = {k[len("blocks.1."):]: v for k, v in full.items() if k.startswith("blocks.1.")}
    x = O.synth_tensor("sbblock.x", (B, T, C), 51).requires_grad_(True)       # channels-last like the oracle
    up = O.synth_tensor("sbblock.up", (B, T, C), 52)
    if sbm is not None:
        blk = sbm.SERes2NetBlock(C, C, res2net_scale=cfg.res2net_scale, se_channels=cfg.se_channels,
                                 kernel_size=cfg.kernel_sizes[1], dilation=dil)
        load_params(blk, sd)
        blk.train()
        y = blk(x.transpose(1, 2)).transpose(1, 2)
        (y * up).sum().backward()
        grads = {n: p.grad for n, p in blk.named_parameters()}
    else:
        sdg = {"blocks.1." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
        y = E.se_res2net_block(x, sdg, "blocks.1.", cfg, dil)
        (y * up).sum().backward()
        grads = {k[len("blocks.1."):]: v.grad for k, v in sdg.items()}
    g = {"x": x.detach(), "upstream": up, "out": y, "dx": x.grad}
    for n, gr in grads.items():
        g["grad." + n] = gr
    np.savez_compressed(os.path.join(out, "g15_sb_seres2net.npz"), **to_np(g), **meta)
    print("g15_sb_seres2net: out norm", float(y.norm()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--self-test", metavar="DIR", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    sbm, ver, path = get_classes(a.self_test is not None)
    out = a.self_test or os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    meta = {"sb_version": np.array(str(ver)), "sb_file": np.array(str(path))}
    golden_asp(sbm, meta, out)
    golden_ecapa_tiny(sbm, meta, out)
    golden_seres2net(sbm, meta, out)
    print("written to", out, "| speechbrain", ver)


if __name__ == "__main__":
    main()
