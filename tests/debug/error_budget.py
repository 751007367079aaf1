"""Error budget of reduced-precision storage on the base model (CPU simulation, B=2, 3 s).

Re-runs the oracle's forward with a rounding function applied at the points where the engine stores an
activation / reads a GEMM operand, for several storage policies, and prints the embedding rel-L2 against the
f32 oracle per policy (and per layer).  Used to choose the precision policy of the benchmarked mode
(embeddings within 1e-3 rel-L2).

    python tests/debug/error_budget.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import w2v2_oracle as O  # noqa: E402

import torch.nn.functional as F  # noqa: E402


def rnd(dt):
    if dt is None:
        return lambda x: x
    return lambda x: x.to(dt).to(torch.float32)


def run(wav, sd, cfg, *, op, res, conv, attn_p=None, layers=None):
    """op: dtype of GEMM operands (activations + weights); res: dtype of the residual stream / LN inputs+outputs
    (None = f32); conv: dtype of the conv-stack activations."""
    qo, qr, qc = rnd(op), rnd(res), rnd(conv)
    qp = rnd(attn_p if attn_p is not None else op)
    W = lambda n: qo(sd[n])
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
            h = F.conv1d(h, qc(w), None, stride=s)
        h = qc(O.gelu(h))
    feat = h.transpose(1, 2)
    n = qo(O.layer_norm(feat, sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], 1e-5))
    h0 = qr(n @ W("feature_projection.projection.weight").t() + sd["feature_projection.projection.bias"])
    # pos conv: operand copy of h0
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(qo(h0).transpose(1, 2), qo(O.pos_conv_weight(sd)), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=G)
    y = y[:, :, :-1]
    pos = qr(O.gelu(y).transpose(1, 2))
    x = qr(O.layer_norm(h0 + pos, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5))
    outs = []
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    B, T, _ = x.shape
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        xo = qo(x)
        qkv = [qo(xo @ W(p + f"attention.{n}.weight").t() + sd[p + f"attention.{n}.bias"]).view(B, T, nh, d).transpose(1, 2)
               for n in ("q_proj", "k_proj", "v_proj")]
        pr = torch.softmax((qkv[0] @ qkv[1].transpose(2, 3)) * d ** -0.5, dim=-1)
        ctx = qo((qp(pr) @ qkv[2]).transpose(1, 2).reshape(B, T, H))
        a = qr(ctx @ W(p + "attention.out_proj.weight").t() + sd[p + "attention.out_proj.bias"])
        x1 = qr(O.layer_norm(x + a, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5))
        hh = qo(O.gelu(qo(x1) @ W(p + "feed_forward.intermediate_dense.weight").t() + sd[p + "feed_forward.intermediate_dense.bias"]))
        f = qr(hh @ W(p + "feed_forward.output_dense.weight").t() + sd[p + "feed_forward.output_dense.bias"])
        x = qr(O.layer_norm(x1 + f, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5))
        outs.append(x)
    return O.mean_std_pool(x), outs


def main():
    torch.set_num_threads(8)
    cfg = O.OracleConfig.base()
    sd = O.make_state_dict(cfg, 20211)
    wav, _ = O.synth_batch(2, 48000, 10, seed=5)
    wav = wav[:, 0, :]
    bf, fp = torch.bfloat16, torch.float16
    with torch.no_grad():
        ref, rl = run(wav, sd, cfg, op=None, res=None, conv=None)
        pol = {
            "all bf16 (round 1)": dict(op=bf, res=bf, conv=bf),
            "bf16 operands, f32 residual stream": dict(op=bf, res=None, conv=bf),
            "bf16 operands, f32 residual, f32 conv": dict(op=bf, res=None, conv=None),
            "all fp16": dict(op=fp, res=fp, conv=fp),
            "fp16 operands, f32 residual stream": dict(op=fp, res=None, conv=fp),
            "fp16 operands, f32 residual, bf16 conv": dict(op=fp, res=None, conv=bf),
            "fp16 enc operands, bf16 residual": dict(op=fp, res=bf, conv=fp),
        }
        for name, kw in pol.items():
            e, ol = run(wav, sd, cfg, **kw)
            err = float((e - ref).norm() / ref.norm())
            per = [float((a - b).norm() / b.norm()) for a, b in zip(ol, rl)]
            print(f"{name:45s} emb rel-L2 {err:.2e}   layers " + " ".join(f"{v:.1e}" for v in per[::3] + per[-1:]))


if __name__ == "__main__":
    main()
