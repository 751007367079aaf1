"""Ablation of the error budget: all-16-bit storage, then one storage point at a time kept exact (f32).
    python tests/debug/error_budget2.py [fp16|bf16]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import w2v2_oracle as O  # noqa: E402


def run(wav, sd, cfg, dt, exact=(), upto=None):
    def q(name):
        if dt is None or name in exact or any(name.startswith(e) for e in exact if e.endswith("*") and name.startswith(e[:-1])):
            return lambda x: x
        return lambda x: x.to(dt).to(torch.float32)
    h = wav[:, None, :]
    for i, s in enumerate(cfg.conv_stride):
        w = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
        if i == 0:
            h = F.conv1d(h, w, None, stride=s)
            mu = h.mean(dim=2, keepdim=True)
            var = h.var(dim=2, unbiased=False, keepdim=True)
            h = (h - mu) / torch.sqrt(var + 1e-5) * sd["feature_extractor.conv_layers.0.layer_norm.weight"][None, :, None] \
                + sd["feature_extractor.conv_layers.0.layer_norm.bias"][None, :, None]
        else:
            h = F.conv1d(h, q(f"convw{i}")(w), None, stride=s)
        h = q(f"conv{i}")(O.gelu(h))
    feat = h.transpose(1, 2)
    n = q("ln_feat")(O.layer_norm(feat, sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], 1e-5))
    h0 = q("h0")(n @ q("projw")(sd["feature_projection.projection.weight"]).t() + sd["feature_projection.projection.bias"])
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(h0.transpose(1, 2), q("posw")(O.pos_conv_weight(sd)), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=G)
    y = y[:, :, :-1]
    pos = q("pos")(O.gelu(y).transpose(1, 2))
    x = q("x0")(O.layer_norm(h0 + pos, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5))
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    B, T, _ = x.shape
    L = cfg.num_hidden_layers if upto is None else upto
    for l in range(L):
        p = f"encoder.layers.{l}."
        W = lambda n: q("encw")(sd[n])
        qkv = [q("qkv")(x @ W(p + f"attention.{n}.weight").t() + sd[p + f"attention.{n}.bias"]).view(B, T, nh, d).transpose(1, 2)
               for n in ("q_proj", "k_proj", "v_proj")]
        pr = torch.softmax((qkv[0] @ qkv[1].transpose(2, 3)) * d ** -0.5, dim=-1)
        ctx = q("ctx")((q("P")(pr) @ qkv[2]).transpose(1, 2).reshape(B, T, H))
        a = q("a")(ctx @ W(p + "attention.out_proj.weight").t() + sd[p + "attention.out_proj.bias"])
        x1 = q("x1")(O.layer_norm(x + a, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5))
        hh = q("h")(O.gelu(x1 @ W(p + "feed_forward.intermediate_dense.weight").t() + sd[p + "feed_forward.intermediate_dense.bias"]))
        f = q("f")(hh @ W(p + "feed_forward.output_dense.weight").t() + sd[p + "feed_forward.output_dense.bias"])
        x = q("x")(O.layer_norm(x1 + f, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5))
    return O.mean_std_pool(x), x


def main():
    torch.set_num_threads(8)
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
    cfg = O.OracleConfig.base()
    sd = O.make_state_dict(cfg, 20211)
    wav, _ = O.synth_batch(2, 48000, 10, seed=5)
    wav = wav[:, 0, :]
    rel = lambda a, b: float((a - b).norm() / b.norm())
    with torch.no_grad():
        ref, xr = run(wav, sd, cfg, None)
        e, x = run(wav, sd, cfg, dt)
        print(f"all {dt}: emb {rel(e, ref):.3e}  last hidden {rel(x, xr):.3e}")
        convs = tuple(f"conv{i}" for i in range(7)) + tuple(f"convw{i}" for i in range(1, 7))
        groups = {
            "conv stack acts+weights": convs,
            "conv acts only": tuple(f"conv{i}" for i in range(7)),
            "conv weights only": tuple(f"convw{i}" for i in range(1, 7)),
            "conv layers 3-6 (acts+w)": tuple(f"conv{i}" for i in range(3, 7)) + tuple(f"convw{i}" for i in range(3, 7)),
            "conv layers 4-6 (acts+w)": tuple(f"conv{i}" for i in range(4, 7)) + tuple(f"convw{i}" for i in range(4, 7)),
            "ln_feat+projw+h0": ("ln_feat", "projw", "h0"),
            "posw+pos+x0": ("posw", "pos", "x0"),
            "whole front (conv..x0)": convs + ("ln_feat", "projw", "h0", "posw", "pos", "x0"),
            "encoder weights": ("encw",),
            "qkv": ("qkv",), "P": ("P",), "ctx": ("ctx",), "a": ("a",), "x1": ("x1",), "h": ("h",), "f": ("f",), "x": ("x",),
            "residual stream (a,x1,f,x)": ("a", "x1", "f", "x"),
            "all encoder layer points": ("encw", "qkv", "P", "ctx", "a", "x1", "h", "f", "x"),
        }
        for name, ex in groups.items():
            e, x = run(wav, sd, cfg, dt, exact=ex)
            print(f"  exact {name:32s}: emb {rel(e, ref):.3e}  last hidden {rel(x, xr):.3e}")


if __name__ == "__main__":
    main()
