"""CPU restatement (torch fp32, functional) of the reference hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  This file is our own
restatement of the arithmetic the reference dispatches through HuggingFace
``Wav2Vec2Model`` / torch; it is pinned against the real reference (imported in
the authoring container) by ``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``
(committed fixtures); ``tests/test_oracle_golden.py`` checks this file against them on CPU,
wherever it runs.  Gradients come from torch autograd over these functions.

Citations: ``ref:`` = /root/reference/, ``HF:`` = transformers 5.15.0
``models/wav2vec2/modeling_wav2vec2.py`` (the reference pins ^4.8.2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Shape description of the HF model the reference loads (ref: src/models/wav2vec2.py:37-51)."""

    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    # the "-lv60" / xlsr family of checkpoints: pre-LN encoder (HF:611-654,729-802), a LayerNorm over
    # the channels after EVERY convolution instead of the layer-0 GroupNorm (HF:275-299), convolutions with bias
    do_stable_layer_norm: bool = False
    feat_extract_norm: str = "group"
    conv_bias: bool = False

    @staticmethod
    def base() -> "OracleConfig":
        return OracleConfig()

    @staticmethod
    def large() -> "OracleConfig":
        return OracleConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                            intermediate_size=4096)

    @staticmethod
    def tiny() -> "OracleConfig":
        # SURVEY 8c golden G1
        return OracleConfig(conv_dim=(32,) * 7, hidden_size=64, num_hidden_layers=2,
                            num_attention_heads=4, intermediate_size=128,
                            num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)

    def num_frames(self, n_samples: int) -> int:
        length = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            length = (length - k) // s + 1          # HF:997-1016 _conv_out_length
        return length


# --------------------------------------------------------------------------- weights
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """HF state-dict names/shapes of Wav2Vec2Model (group-norm or layer-norm CNN, post- or pre-LN encoder: the encoder's
    names are the same in both)."""
    shp: Dict[str, Tuple[int, ...]] = {}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        shp[f"feature_extractor.conv_layers.{i}.conv.weight"] = (c, cin, k)
        if cfg.conv_bias:
            shp[f"feature_extractor.conv_layers.{i}.conv.bias"] = (c,)
        if i == 0 or cfg.feat_extract_norm == "layer":
            shp[f"feature_extractor.conv_layers.{i}.layer_norm.weight"] = (c,)
            shp[f"feature_extractor.conv_layers.{i}.layer_norm.bias"] = (c,)
        cin = c
    H, C = cfg.hidden_size, cfg.conv_dim[-1]
    shp["feature_projection.layer_norm.weight"] = (C,)
    shp["feature_projection.layer_norm.bias"] = (C,)
    shp["feature_projection.projection.weight"] = (H, C)
    shp["feature_projection.projection.bias"] = (H,)
    shp["masked_spec_embed"] = (H,)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    shp["encoder.pos_conv_embed.conv.bias"] = (H,)
    shp["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, K)
    shp["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (H, H // G, K)
    shp["encoder.layer_norm.weight"] = (H,)
    shp["encoder.layer_norm.bias"] = (H,)
    I = cfg.intermediate_size
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            shp[p + f"attention.{n}.weight"] = (H, H)
            shp[p + f"attention.{n}.bias"] = (H,)
        shp[p + "layer_norm.weight"] = (H,)
        shp[p + "layer_norm.bias"] = (H,)
        shp[p + "feed_forward.intermediate_dense.weight"] = (I, H)
        shp[p + "feed_forward.intermediate_dense.bias"] = (I,)
        shp[p + "feed_forward.output_dense.weight"] = (H, I)
        shp[p + "feed_forward.output_dense.bias"] = (H,)
        shp[p + "final_layer_norm.weight"] = (H,)
        shp[p + "final_layer_norm.bias"] = (H,)
    return shp


def synth_tensor(name: str, shape: Sequence[int], seed: int) -> Tensor:
    """Deterministic name-keyed PCG64 tensor: both the golden script and the GPU box regenerate identical weights, so
    no 378 MB checkpoint is ever committed.  The generator itself is plain numpy and lives with the other
    synthetic-data helpers of the package (w2v2_speaker_amd/data/synthetic.py: bench.py's EER leg needs the goldens'
    weights and must not import this directory); no arithmetic of the path is shared that way."""
    from w2v2_speaker_amd.data.synthetic import synth_weight
    return torch.from_numpy(synth_weight(name, shape, seed))


def make_state_dict(cfg: OracleConfig, seed: int = 20211) -> StateDict:
    return {n: synth_tensor(n, s, seed) for n, s in param_shapes(cfg).items()}


# --------------------------------------------------------------------------- input
def normalise_waveform(x: Tensor) -> Tensor:
    """ref: src/data/preprocess/input_normalisation.py:54-67 -- whole-utterance
    (x-mean)/(std+1e-5) with the unbiased std; applied per utterance ([.., N])."""
    mean = x.mean(dim=-1, keepdim=True)
    std = x.std(dim=-1, keepdim=True)           # unbiased
    return (x - mean) / (std + 1e-5)


def synth_batch(batch: int, n_samples: int, num_speakers: int, seed: int = 42133724):
    """Synthetic workload of SURVEY 8(d): N(0,1) waveform, normalised per utterance, [B,1,N]."""
    g = np.random.Generator(np.random.PCG64(seed))
    wav = torch.from_numpy(g.standard_normal((batch, n_samples)).astype(np.float32))
    wav = normalise_waveform(wav)[:, None, :]
    label = torch.from_numpy(g.integers(0, num_speakers, size=(batch,)).astype(np.int64))
    return wav, label


# --------------------------------------------------------------------------- wav2vec2 forward
def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))      # ACT2FN["gelu"], exact erf


def feature_extractor(x: Tensor, sd: StateDict, cfg: OracleConfig) -> Tensor:
    """HF:382-419 (Wav2Vec2FeatureEncoder): x [B,N] -> [B,C,T]; layer 0 = conv + GroupNorm(C groups)
    + GELU (HF:302-323), layers 1-6 = conv + GELU (HF:254-272); no conv bias."""
    h = x[:, None, :]
    for i, s in enumerate(cfg.conv_stride):
        w = sd[f"feature_extractor.conv_layers.{i}.conv.weight"]
        h = F.conv1d(h, w, sd[f"feature_extractor.conv_layers.{i}.conv.bias"] if cfg.conv_bias else None, stride=s)
        if cfg.feat_extract_norm == "layer":
            # HF:275-299 (Wav2Vec2LayerNormConvLayer): nn.LayerNorm(C) over the channels of every frame, default eps 1e-5
            h = layer_norm(h.transpose(1, 2), sd[f"feature_extractor.conv_layers.{i}.layer_norm.weight"],
                           sd[f"feature_extractor.conv_layers.{i}.layer_norm.bias"], 1e-5).transpose(1, 2)
        elif i == 0:
            gamma = sd["feature_extractor.conv_layers.0.layer_norm.weight"]
            beta = sd["feature_extractor.conv_layers.0.layer_norm.bias"]
            mu = h.mean(dim=2, keepdim=True)
            var = h.var(dim=2, unbiased=False, keepdim=True)
            h = (h - mu) / torch.sqrt(var + 1e-5) * gamma[None, :, None] + beta[None, :, None]
        h = gelu(h)
    return h


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def feature_projection(feat_btc: Tensor, sd: StateDict, cfg: OracleConfig) -> Tensor:
    """HF:422-435: LayerNorm(C) -> Linear(C->H); dropout is the caller's business (p=0 here)."""
    n = layer_norm(feat_btc, sd["feature_projection.layer_norm.weight"],
                   sd["feature_projection.layer_norm.bias"], cfg.layer_norm_eps)
    return n @ sd["feature_projection.projection.weight"].t() + sd["feature_projection.projection.bias"]


def apply_time_mask(h: Tensor, mask: Optional[Tensor], sd: StateDict) -> Tensor:
    """HF:1290-1292: h[mask] = masked_spec_embed (mask [B,T] bool)."""
    if mask is None:
        return h
    return torch.where(mask[:, :, None], sd["masked_spec_embed"][None, None, :], h)


def pos_conv_weight(sd: StateDict) -> Tensor:
    """weight_norm(dim=2): w = g * v / ||v||, norm over dims (0,1) per tap (HF:341-349)."""
    g = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return g * v / torch.sqrt((v * v).sum(dim=(0, 1), keepdim=True))


def pos_conv_embed(h: Tensor, sd: StateDict, cfg: OracleConfig) -> Tensor:
    """HF:326-379: grouped Conv1d(H,H,K,pad=K//2,groups=G)+bias on [B,H,T], drop the last frame
    when K is even, GELU; returns [B,T,H]."""
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"],
                 padding=K // 2, groups=G)
    if K % 2 == 0:
        y = y[:, :, :-1]
    return gelu(y).transpose(1, 2)


def attention(x: Tensor, sd: StateDict, prefix: str, n_heads: int) -> Tensor:
    """HF:466-548 + eager_attention_forward HF:438-463; no attention mask on this path."""
    B, T, H = x.shape
    d = H // n_heads

    def proj(n):
        return (x @ sd[prefix + f"attention.{n}.weight"].t() + sd[prefix + f"attention.{n}.bias"]) \
            .view(B, T, n_heads, d).transpose(1, 2)

    q, k, v = proj("q_proj"), proj("k_proj"), proj("v_proj")
    p = torch.softmax((q @ k.transpose(2, 3)) * (d ** -0.5), dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B, T, H)
    return ctx @ sd[prefix + "attention.out_proj.weight"].t() + sd[prefix + "attention.out_proj.bias"]


def encoder_layer(x: Tensor, sd: StateDict, l: int, cfg: OracleConfig) -> Tensor:
    """HF:575-608 post-LN block: x = LN1(x + Attn(x)); x = LN2(x + FFN(x))."""
    p = f"encoder.layers.{l}."
    x = layer_norm(x + attention(x, sd, p, cfg.num_attention_heads),
                   sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.layer_norm_eps)
    f = gelu(x @ sd[p + "feed_forward.intermediate_dense.weight"].t()
             + sd[p + "feed_forward.intermediate_dense.bias"])
    f = f @ sd[p + "feed_forward.output_dense.weight"].t() + sd[p + "feed_forward.output_dense.bias"]
    return layer_norm(x + f, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"],
                      cfg.layer_norm_eps)


def encoder_layer_stable(x: Tensor, sd: StateDict, l: int, cfg: OracleConfig) -> Tensor:
    """HF:611-654 pre-LN block (do_stable_layer_norm): x = x + Attn(LN1(x)); x = x + FFN(LN2(x))."""
    p = f"encoder.layers.{l}."
    x = x + attention(layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], cfg.layer_norm_eps), sd, p,
                      cfg.num_attention_heads)
    n = layer_norm(x, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], cfg.layer_norm_eps)
    f = gelu(n @ sd[p + "feed_forward.intermediate_dense.weight"].t() + sd[p + "feed_forward.intermediate_dense.bias"])
    return x + f @ sd[p + "feed_forward.output_dense.weight"].t() + sd[p + "feed_forward.output_dense.bias"]


def encoder_stable(h: Tensor, sd: StateDict, cfg: OracleConfig, skip_layers: Sequence[int] = (),
                   return_stages: bool = False):
    """HF:729-802 (Wav2Vec2EncoderStableLayerNorm): x = x + posconv(x); pre-LN layers; ONE LayerNorm at the end.
    Stage l is the un-normalised residual stream after layer l, as HF's hidden_states has it."""
    stages = {}
    pos = pos_conv_embed(h, sd, cfg)
    x = h + pos
    if return_stages:
        stages["pos_conv"] = pos
        stages["enc_in"] = x
    for l in range(cfg.num_hidden_layers):
        if l in skip_layers:
            continue
        x = encoder_layer_stable(x, sd, l, cfg)
        if return_stages:
            stages[f"layer{l}"] = x
    x = layer_norm(x, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], cfg.layer_norm_eps)
    return (x, stages) if return_stages else x


def encoder(h: Tensor, sd: StateDict, cfg: OracleConfig, skip_layers: Sequence[int] = (),
            return_stages: bool = False):
    """HF:657-726: x = LN(x + posconv(x)); 12/24 layers; LayerDrop is injected via skip_layers."""
    if cfg.do_stable_layer_norm:
        return encoder_stable(h, sd, cfg, skip_layers, return_stages)
    stages = {}
    pos = pos_conv_embed(h, sd, cfg)
    x = layer_norm(h + pos, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"],
                   cfg.layer_norm_eps)
    if return_stages:
        stages["pos_conv"] = pos
        stages["enc_in"] = x
    for l in range(cfg.num_hidden_layers):
        if l in skip_layers:
            continue
        x = encoder_layer(x, sd, l, cfg)
        if return_stages:
            stages[f"layer{l}"] = x
    return (x, stages) if return_stages else x


def wav2vec2_forward(wav: Tensor, sd: StateDict, cfg: OracleConfig,
                     mask_time_indices: Optional[Tensor] = None,
                     insert_cls_token: bool = False, cls_token_constant: float = 1.0,
                     skip_layers: Sequence[int] = (), return_stages: bool = False):
    """ref: src/models/wav2vec2.py:126-146 (+ :62-76) minus the two cancelling transposes:
    wav [B,N] -> last_hidden_state [B,T,H] (the wrapper returns its transpose [B,H,T]).
    Dropouts are 0 (parity mode); time mask / LayerDrop decisions are injected."""
    stages = {}
    feat = feature_extractor(wav, sd, cfg)                      # [B,C,T]
    h = feature_projection(feat.transpose(1, 2), sd, cfg)       # [B,T,H]
    if return_stages:
        stages["conv_out"] = feat.transpose(1, 2)
        stages["proj"] = h
    if insert_cls_token:
        # ref: src/models/wav2vec2.py:128-140 -- constant token prepended, SpecAugment bypassed
        cls = torch.full((h.shape[0], 1, h.shape[2]), float(cls_token_constant), dtype=h.dtype)
        h = torch.cat([cls, h], dim=1)
    else:
        h = apply_time_mask(h, mask_time_indices, sd)
    out = encoder(h, sd, cfg, skip_layers, return_stages)
    if return_stages:
        out, st = out
        stages.update(st)
        return out, stages
    return out


# --------------------------------------------------------------------------- pooling
def mean_std_pool(x_bth: Tensor) -> Tensor:
    """ref: src/layers/pooling.py:43-44 -- cat(std_unbiased, mean) over time (std FIRST)."""
    std, mean = torch.std_mean(x_bth, dim=1)
    return torch.cat([std, mean], dim=1)


def mean_pool(x_bth: Tensor) -> Tensor:
    """ref: src/layers/pooling.py:29-30"""
    return x_bth.mean(dim=1)


def max_pool(x_bth: Tensor) -> Tensor:
    """ref: src/layers/pooling.py:79-80"""
    return x_bth.max(dim=1).values


def index_pool(x_bth: Tensor, method: str) -> Tensor:
    """ref: src/layers/pooling.py:118-136 -- NB "middle" returns the LAST frame (quirk Q2)."""
    if method in ("first", "first+cls"):
        return x_bth[:, 0, :].clone()
    if method in ("middle", "last"):
        return x_bth[:, -1, :].clone()
    raise ValueError(method)


def quantile_pool(x_bth: Tensor) -> Tensor:
    """ref: src/layers/pooling.py:57-67"""
    q = torch.quantile(x_bth, torch.tensor([0, 0.25, 0.5, 0.75, 1.0]), dim=1)
    return torch.flatten(q.transpose(0, 1), 1, 2)


def attentive_stat_pool(x_bth: Tensor, asp: StateDict, bn_eps: float = 1e-5) -> Tensor:
    """speechbrain 0.5.x AttentiveStatisticsPooling(C, attention_channels=128, global_context=True)
    (NOT in /root/reference, NOT installed -> restated from the published definition, SURVEY App. D;
    parity unpinned).  ref call site: src/layers/pooling.py:87-106.  BatchNorm uses batch statistics
    (training mode).  Output [mean, std] (mean FIRST, opposite of mean_std_pool)."""
    x = x_bth.transpose(1, 2)                                    # [B,C,T]
    B, C, T = x.shape
    eps = 1e-12
    mean = x.mean(dim=2, keepdim=True)
    std = torch.sqrt(((x - mean) ** 2).mean(dim=2, keepdim=True).clamp(eps))
    attn_in = torch.cat([x, mean.expand(-1, -1, T), std.expand(-1, -1, T)], dim=1)     # [B,3C,T]
    a = F.conv1d(attn_in, asp["tdnn.conv.weight"], asp["tdnn.conv.bias"])              # k=1
    a = F.relu(a)
    mu = a.mean(dim=(0, 2), keepdim=True)
    var = a.var(dim=(0, 2), unbiased=False, keepdim=True)
    a = (a - mu) / torch.sqrt(var + bn_eps) * asp["tdnn.norm.weight"][None, :, None] \
        + asp["tdnn.norm.bias"][None, :, None]
    a = torch.tanh(a)
    a = F.conv1d(a, asp["conv.weight"], asp["conv.bias"])                              # [B,C,T]
    w = torch.softmax(a, dim=2)
    wmean = (w * x).sum(dim=2)
    wstd = torch.sqrt(((w * (x - wmean[:, :, None]) ** 2).sum(dim=2)).clamp(eps))
    return torch.cat([wmean, wstd], dim=1)


# --------------------------------------------------------------------------- heads
def aam_softmax(x: Tensor, fc_weights: Tensor, label: Tensor, margin: float = 0.2,
                scale: float = 30.0, easy_margin: bool = False) -> Tuple[Tensor, Tensor]:
    """ref: src/optim/loss/aam_softmax.py:50-74 (both branches of easy_margin, :60-63) -> (loss, softmax[B,C])."""
    cos_m, sin_m = math.cos(margin), math.sin(margin)
    th = math.cos(math.pi - margin)
    mm = math.sin(math.pi - margin) * margin
    xn = x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)               # F.normalize
    wn = fc_weights / fc_weights.norm(dim=1, keepdim=True).clamp_min(1e-12)
    cosine = xn @ wn.t()
    sine = torch.sqrt((1.0 - cosine * cosine).clamp(0, 1))
    phi = cosine * cos_m - sine * sin_m
    phi = torch.where(cosine > 0, phi, cosine) if easy_margin else torch.where((cosine - th) > 0, phi, cosine - mm)
    one_hot = torch.zeros_like(cosine)
    one_hot.scatter_(1, label.view(-1, 1), 1)
    output = (one_hot * phi + (1.0 - one_hot) * cosine) * scale
    logp = torch.log_softmax(output, dim=1)
    loss = -logp.gather(1, label.view(-1, 1)).mean()
    return loss, torch.softmax(output, dim=1)


def ce_head(x: Tensor, weight: Tensor, bias: Tensor, label: Tensor) -> Tuple[Tensor, Tensor]:
    """ref: src/lightning_modules/speaker/wav2vec2_fc.py:199-210 (last nn.Linear) +
    src/optim/loss/cross_entropy.py:27-31 -> (loss, softmax)."""
    logits = x @ weight.t() + bias
    logp = torch.log_softmax(logits, dim=1)
    loss = -logp.gather(1, label.view(-1, 1)).mean()
    return loss, torch.softmax(logits, dim=1)


def speaker_embedding(wav_b1n: Tensor, sd: StateDict, cfg: OracleConfig, pooling: str = "mean+std",
                      mask_time_indices: Optional[Tensor] = None, skip_layers: Sequence[int] = ()):
    """ref: src/lightning_modules/speaker/wav2vec2_fc.py:414-431 + :363-385 (masker = identity, Q3)."""
    wav = wav_b1n[:, 0, :] if wav_b1n.dim() == 3 else wav_b1n
    cls = pooling == "first+cls"
    h = wav2vec2_forward(wav, sd, cfg, mask_time_indices=mask_time_indices, insert_cls_token=cls,
                         skip_layers=skip_layers)
    if pooling == "mean+std":
        return mean_std_pool(h)
    if pooling == "mean":
        return mean_pool(h)
    if pooling == "max":
        return max_pool(h)
    if pooling == "quantile":
        return quantile_pool(h)
    return index_pool(h, pooling)


def paired_equality_scores(wav_left: Tensor, wav_right: Tensor, sd: StateDict, cfg: OracleConfig, lin_w: Tensor,
                           lin_b: Tensor, cls_token_constant: float = 1.0, sep_token_constant: float = -1.0,
                           skip_layers: Sequence[int] = ()) -> Tensor:
    """ref: src/lightning_modules/speaker/wav2vec2_paired_input.py:163-207 (compute_speaker_equality): both waveforms
    through the conv stack + projection, sequence [CLS] left [SEP] right [SEP] (constant tokens) through the encoder
    (no SpecAugment on this path), Linear(H, 1) on token 0 -> logits [B, 1]."""
    f1 = feature_projection(feature_extractor(wav_left, sd, cfg).transpose(1, 2), sd, cfg)
    f2 = feature_projection(feature_extractor(wav_right, sd, cfg).transpose(1, 2), sd, cfg)
    B, _, H = f1.shape
    tok = lambda c: torch.full((B, 1, H), float(c), dtype=f1.dtype)
    seq = torch.cat([tok(cls_token_constant), f1, tok(sep_token_constant), f2, tok(sep_token_constant)], dim=1)
    out = encoder(seq, sd, cfg, skip_layers)
    return out[:, 0, :] @ lin_w.t() + lin_b


def bce_with_logits(logits: Tensor, label: Tensor) -> Tuple[Tensor, Tensor]:
    """ref: src/optim/loss/binary_cross_entropy.py:24-40 -> (mean loss, sigmoid prediction)."""
    lg = logits.squeeze().to(torch.float32)
    y = label.squeeze().to(torch.float32)
    return F.binary_cross_entropy_with_logits(lg, y), torch.sigmoid(lg).detach()


# --------------------------------------------------------------------------- optimiser
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float,
              beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam (no amsgrad, wd 0) single-tensor update, in place
    (ref: config/optim/algo/adam.yaml:1-16, wired at src/main.py:323-335)."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def one_cycle(step: int, total_steps: int, max_lr: float, pct_start: float = 0.3,
              div_factor: float = 25.0, final_div_factor: float = 1e4,
              base_momentum: float = 0.85, max_momentum: float = 0.95) -> Tuple[float, float]:
    """torch OneCycleLR (cos anneal, two phases, cycle_momentum=True) -> (lr, beta1) to be used for
    optimiser step number ``step`` (0-based)  (ref: config/optim/schedule/one_cycle.yaml:3-20)."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1
    end2 = total_steps - 1

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    if step <= end1:
        pct = step / end1
        return cos(initial_lr, max_lr, pct), cos(max_momentum, base_momentum, pct)
    pct = (step - end1) / (end2 - end1)
    return cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)


# --------------------------------------------------------------------------- evaluation
def cosine_scores(a: Tensor, b: Tensor) -> Tensor:
    """ref: src/evaluation/speaker/cosine_distance.py:237-243 + speaker_recognition_evaluator.py:81
    ((s+1)/2 clipped to [0,1])."""
    s = F.cosine_similarity(a, b, dim=1)
    return torch.clip((s + 1) / 2, 0, 1)


def roc_curve(y_true: np.ndarray, y_score: np.ndarray, pos_label=1):
    """Restatement of sklearn.metrics.roc_curve (drop_intermediate=True) as used by
    ref: src/eval_metrics.py:54-79."""
    y_true = (np.asarray(y_true) == pos_label)
    y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, y_true = y_score[order], y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    thr = y_score[idx]
    if len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    thr = np.r_[np.inf, thr]
    return fps / fps[-1], tps / tps[-1], thr


def calculate_eer(groundtruth, scores, pos_label: int = 1) -> Tuple[float, float]:
    """ref: src/eval_metrics.py:54-79 -- EER = root of 1 - x - interp(fpr,tpr)(x)."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq

    fpr, tpr, thresholds = roc_curve(groundtruth, scores, pos_label)
    eer = brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0)
    thresh = interp1d(fpr, thresholds)(eer)
    return float(eer), float(thresh)


def calculate_mdc(groundtruth, scores, c_miss=1.0, c_fa=1.0, p_target=0.05,
                  pos_label: int = 1) -> Tuple[float, float]:
    """ref: src/eval_metrics.py:90-206 -- thresholds = scores sorted ascending (stable);
    fnr[i] = #targets with score <= thr[i] / #targets; fpr[i] = 1 - #non-targets <= thr[i] / #non;
    min over i (first minimum) of c_miss*fnr*p + c_fa*fpr*(1-p), normalised by c_def."""
    gt = np.asarray(groundtruth, dtype=np.float64)
    sc = np.asarray(scores, dtype=np.float64)
    order = np.argsort(sc, kind="stable")
    gt, thr = gt[order], sc[order]
    fnr = np.cumsum(gt) / gt.sum()
    fpr = 1.0 - np.cumsum(1.0 - gt) / (len(gt) - gt.sum())
    c = c_miss * fnr * p_target + c_fa * fpr * (1 - p_target)
    i = int(np.argmin(c))                     # first minimum, like the strict '<' sweep
    c_def = min(c_miss * p_target, c_fa * (1 - p_target))
    return float(c[i] / c_def), float(thr[i])
