"""Round-6 question: can the second K pass of the two-term value / output projections (W = hi + lo, +3.5-5 % step time) be replaced
by a MEAN-FIELD correction -- out += mean_t(operand[b]) @ lo^T, one row per utterance -- which captures exactly the part of the weight
rounding error that mean pooling does not average out?  CPU simulation on the base model like error_budget.py (fp16 operands and
storage, f32 accumulation); prints the embedding rel-L2 against the f32 oracle per policy.

    python tests/debug/error_budget_meanfield.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import w2v2_oracle as O  # noqa: E402
import torch.nn.functional as F  # noqa: E402

fp = torch.float16
q = lambda x: x.to(fp).to(torch.float32)


def run(wav, sd, cfg, mode):
    """mode: 'f32' | 'none' (single-term fp16 weights everywhere) | 'full' (exact value / output projection weights = two-term)
    | 'mean' (single-term + per-utterance mean-field correction on those two products)."""
    lowp = mode != "f32"
    qo = q if lowp else (lambda x: x)
    W = lambda n: qo(sd[n])

    def proj(xin, name, two):
        w = sd[name]
        if not lowp or not two or mode == "none":
            return xin @ W(name).t()
        if mode == "full":
            return xin @ w.t()                                   # hi + lo = the f32 weight (to 2^-22)
        lo = q(w - q(w))                                         # the residual plane as stored (fp16)
        corr = qo(xin.mean(dim=1, keepdim=True)) @ lo.t()        # [B, 1, N]: one row per utterance
        return xin @ W(name).t() + corr

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
            h = F.conv1d(h, qo(w), None, stride=s)
        h = qo(O.gelu(h))
    feat = h.transpose(1, 2)
    n = qo(O.layer_norm(feat, sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], 1e-5))
    h0 = qo(n @ W("feature_projection.projection.weight").t() + sd["feature_projection.projection.bias"])
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(h0.transpose(1, 2), qo(O.pos_conv_weight(sd)), sd["encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=G)[:, :, :-1]
    pos = qo(O.gelu(y).transpose(1, 2))
    x = qo(O.layer_norm(h0 + pos, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5))
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    B, T, _ = x.shape
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        qkv = [qo(proj(x, p + f"attention.{nm}.weight", nm == "v_proj") + sd[p + f"attention.{nm}.bias"]).view(B, T, nh, d).transpose(1, 2)
               for nm in ("q_proj", "k_proj", "v_proj")]
        pr = torch.softmax((qkv[0] @ qkv[1].transpose(2, 3)) * d ** -0.5, dim=-1)
        ctx = qo((qo(pr) @ qkv[2]).transpose(1, 2).reshape(B, T, H))
        a = qo(proj(ctx, p + "attention.out_proj.weight", True) + sd[p + "attention.out_proj.bias"])
        s1 = x + a
        x1 = qo(O.layer_norm(s1, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5))      # statistics from the unrounded sum
        hh = qo(O.gelu(x1 @ W(p + "feed_forward.intermediate_dense.weight").t() + sd[p + "feed_forward.intermediate_dense.bias"]))
        f = qo(hh @ W(p + "feed_forward.output_dense.weight").t() + sd[p + "feed_forward.output_dense.bias"])
        x = qo(O.layer_norm(x1 + f, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5))
    return O.mean_std_pool(x)


def main():
    torch.set_num_threads(8)
    cfg = O.OracleConfig.base()
    for wseed, xseed in ((20211, 42133724), (777, 31337)):
        sd = O.make_state_dict(cfg, wseed)
        wav, _ = O.synth_batch(2, 48000, 10, seed=xseed)
        wav = wav[:, 0, :]
        with torch.no_grad():
            ref = run(wav, sd, cfg, "f32")
            for mode in ("none", "full", "mean"):
                e = run(wav, sd, cfg, mode)
                per = (e - ref).norm(dim=1) / ref.norm(dim=1)
                print(f"weights {wseed}: {mode:5s} per-utterance rel-L2 " + " ".join(f"{float(v):.2e}" for v in per))


if __name__ == "__main__":
    main()
