"""CPU restatement of speechbrain 0.5.x ``ECAPA_TDNN`` as the reference instantiates it
(ref: src/lightning_modules/speaker/ecapa_tdnn.py:75-85, config/network/ecapa_tdnn.yaml:4-30).

TEST INFRASTRUCTURE ONLY (same rule as w2v2_oracle.py).  speechbrain is neither under /root/reference nor
installed: the arithmetic below follows the published ``speechbrain/lobes/models/ECAPA_TDNN.py`` definition
(TDNNBlock = Conv1d("same", reflect) -> ReLU -> BatchNorm1d; Res2Net with cumulative adds; SE gate; MFA concat of the
SE-Res2Net outputs; attentive statistics pooling with global context; BatchNorm1d; 1x1 ``fc``) -- PARITY UNPINNED:
no reference test, fixture or runnable reference pins it; HIP-vs-oracle agreement is self-consistency only.

Tensors are channels-last here ([B, T, C]); BatchNorm uses batch statistics (training mode), biased variance."""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from .w2v2_oracle import attentive_stat_pool, synth_tensor

Tensor = torch.Tensor
BN_EPS = 1e-5


@dataclasses.dataclass
class EcapaConfig:
    input_size: int = 40
    lin_neurons: int = 192
    channels: Tuple[int, ...] = (1024, 1024, 1024, 1024, 3072)
    kernel_sizes: Tuple[int, ...] = (5, 3, 3, 3, 1)
    dilations: Tuple[int, ...] = (1, 2, 3, 4, 1)
    attention_channels: int = 128
    res2net_scale: int = 8
    se_channels: int = 128

    @staticmethod
    def tiny() -> "EcapaConfig":
        return EcapaConfig(input_size=16, lin_neurons=24, channels=(64, 64, 64, 64, 192), attention_channels=16,
                           res2net_scale=4, se_channels=16)


def param_shapes(cfg: EcapaConfig) -> Dict[str, Tuple[int, ...]]:
    """speechbrain state-dict names (Conv1d wrapper -> ``.conv``, BatchNorm1d wrapper -> ``.norm``)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def tdnn(p, cin, cout, k):
        s[p + "conv.conv.weight"] = (cout, cin, k)
        s[p + "conv.conv.bias"] = (cout,)
        s[p + "norm.norm.weight"] = (cout,)
        s[p + "norm.norm.bias"] = (cout,)
    C = cfg.channels
    tdnn("blocks.0.", cfg.input_size, C[0], cfg.kernel_sizes[0])
    for i in range(1, len(C) - 1):
        p = f"blocks.{i}."
        assert C[i - 1] == C[i], "shortcut conv (in != out channels) is not used by the reference config"
        tdnn(p + "tdnn1.", C[i - 1], C[i], 1)
        w = C[i] // cfg.res2net_scale
        for j in range(cfg.res2net_scale - 1):
            tdnn(p + f"res2net_block.blocks.{j}.", w, w, cfg.kernel_sizes[i])
        tdnn(p + "tdnn2.", C[i], C[i], 1)
        s[p + "se_block.conv1.conv.weight"] = (cfg.se_channels, C[i], 1)
        s[p + "se_block.conv1.conv.bias"] = (cfg.se_channels,)
        s[p + "se_block.conv2.conv.weight"] = (C[i], cfg.se_channels, 1)
        s[p + "se_block.conv2.conv.bias"] = (C[i],)
    tdnn("mfa.", C[-1], C[-1], cfg.kernel_sizes[-1])
    A = cfg.attention_channels
    s["asp.tdnn.conv.conv.weight"] = (A, 3 * C[-1], 1)
    s["asp.tdnn.conv.conv.bias"] = (A,)
    s["asp.tdnn.norm.norm.weight"] = (A,)
    s["asp.tdnn.norm.norm.bias"] = (A,)
    s["asp.conv.conv.weight"] = (C[-1], A, 1)
    s["asp.conv.conv.bias"] = (C[-1],)
    s["asp_bn.norm.weight"] = (2 * C[-1],)
    s["asp_bn.norm.bias"] = (2 * C[-1],)
    s["fc.conv.weight"] = (cfg.lin_neurons, 2 * C[-1], 1)
    s["fc.conv.bias"] = (cfg.lin_neurons,)
    return s


def make_state_dict(cfg: EcapaConfig, seed: int = 20211) -> Dict[str, Tensor]:
    sd = {}
    for n, shp in param_shapes(cfg).items():
        t = synth_tensor("ecapa." + n, shp, seed)
        if n.endswith("norm.weight"):
            t = 1.0 + 0.1 * t
        elif n.endswith("bias"):
            t = 0.1 * t
        else:
            t = t * (1.5 / (shp[1] * shp[2]) ** 0.5)
        sd[n] = t
    return sd


def conv1d_same_reflect(x_btc: Tensor, w: Tensor, b: Tensor, dilation: int) -> Tensor:
    """speechbrain Conv1d(padding="same", padding_mode="reflect"), stride 1: pad d (k-1)/2 each side."""
    k = w.shape[2]
    x = x_btc.transpose(1, 2)
    p = dilation * (k - 1) // 2
    if p > 0:
        x = F.pad(x, (p, p), mode="reflect")
    return F.conv1d(x, w, b, dilation=dilation).transpose(1, 2)


def batchnorm_train(x: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """BatchNorm1d in training mode over every axis but the channel one (batch statistics, biased variance)."""
    dims = tuple(range(x.dim() - 1))
    mu = x.mean(dim=dims, keepdim=True)
    var = x.var(dim=dims, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + BN_EPS) * gamma + beta


def tdnn_block(x: Tensor, sd, p: str, dilation: int) -> Tensor:
    y = conv1d_same_reflect(x, sd[p + "conv.conv.weight"], sd[p + "conv.conv.bias"], dilation)
    return batchnorm_train(F.relu(y), sd[p + "norm.norm.weight"], sd[p + "norm.norm.bias"])


def res2net_block(x: Tensor, sd, p: str, scale: int, dilation: int) -> Tensor:
    ys: List[Tensor] = []
    y_i = None
    for i, x_i in enumerate(torch.chunk(x, scale, dim=2)):
        if i == 0:
            y_i = x_i
        elif i == 1:
            y_i = tdnn_block(x_i, sd, p + f"blocks.{i - 1}.", dilation)
        else:
            y_i = tdnn_block(x_i + y_i, sd, p + f"blocks.{i - 1}.", dilation)
        ys.append(y_i)
    return torch.cat(ys, dim=2)


def se_block(x: Tensor, sd, p: str) -> Tensor:
    s = x.mean(dim=1, keepdim=True)                                   # [B,1,C]
    s = F.relu(s @ sd[p + "conv1.conv.weight"][:, :, 0].t() + sd[p + "conv1.conv.bias"])
    s = torch.sigmoid(s @ sd[p + "conv2.conv.weight"][:, :, 0].t() + sd[p + "conv2.conv.bias"])
    return s * x


def se_res2net_block(x: Tensor, sd, p: str, cfg: EcapaConfig, dilation: int) -> Tensor:
    y = tdnn_block(x, sd, p + "tdnn1.", 1)
    y = res2net_block(y, sd, p + "res2net_block.", cfg.res2net_scale, dilation)
    y = tdnn_block(y, sd, p + "tdnn2.", 1)
    y = se_block(y, sd, p + "se_block.")
    return y + x


def ecapa_forward(feat_btf: Tensor, sd, cfg: EcapaConfig, return_stages: bool = False):
    """[B, T, n_mels] -> [B, lin_neurons]."""
    st = {}
    x = tdnn_block(feat_btf, sd, "blocks.0.", cfg.dilations[0])
    st["block0"] = x
    outs = []
    for i in range(1, len(cfg.channels) - 1):
        x = se_res2net_block(x, sd, f"blocks.{i}.", cfg, cfg.dilations[i])
        st[f"block{i}"] = x
        outs.append(x)
    x = torch.cat(outs, dim=2)
    x = tdnn_block(x, sd, "mfa.", cfg.dilations[-1])
    st["mfa"] = x
    asp = {"tdnn.conv.weight": sd["asp.tdnn.conv.conv.weight"], "tdnn.conv.bias": sd["asp.tdnn.conv.conv.bias"],
           "tdnn.norm.weight": sd["asp.tdnn.norm.norm.weight"], "tdnn.norm.bias": sd["asp.tdnn.norm.norm.bias"],
           "conv.weight": sd["asp.conv.conv.weight"], "conv.bias": sd["asp.conv.conv.bias"]}
    x = attentive_stat_pool(x, asp)                                   # [B, 2C]
    st["asp"] = x
    x = batchnorm_train(x, sd["asp_bn.norm.weight"], sd["asp_bn.norm.bias"])
    x = x @ sd["fc.conv.weight"][:, :, 0].t() + sd["fc.conv.bias"]
    return (x, st) if return_stages else x
