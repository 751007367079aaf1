"""Mirror of ref: src/layers/pooling.py on the HIP pooling kernels (csrc/pool.hip).

Inputs follow the reference's call site (``[BATCH, TIME, FEATURE]`` with ``dim_to_reduce=1``,
ref: src/lightning_modules/speaker/wav2vec2_fc.py:172-182); ``dim_to_reduce=2`` takes ``[B, F, T]``."""
from __future__ import annotations

import random

import torch

from .. import ops


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        B, T, H = x.shape
        out = torch.empty(B, (2 * H) if mode == 0 else H, dtype=torch.float32, device=x.device)
        ops.pool_fwd(x, out, mode)
        ctx.save_for_backward(x, out)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.pool_bwd(x, out, dout.contiguous().float(), dx, ctx.mode)
        return dx, None


def _btf(tensor: torch.Tensor, dim_to_reduce: int) -> torch.Tensor:
    if dim_to_reduce == 1:
        return tensor
    if dim_to_reduce == 2:
        return tensor.transpose(1, 2)
    raise ValueError("can only pool dimension 1 or 2")


class MeanStatPool1D(torch.nn.Module):
    def __init__(self, dim_to_reduce: int = 2):
        super().__init__()
        self.dim_to_reduce = dim_to_reduce

    def forward(self, tensor: torch.Tensor):
        return _PoolFn.apply(_btf(tensor, self.dim_to_reduce), ops.POOL_MODES["mean"])


class MeanStdStatPool1D(torch.nn.Module):
    """cat(std_unbiased, mean) -- std FIRST (ref: src/layers/pooling.py:43-44)."""

    def __init__(self, dim_to_reduce: int = 2):
        super().__init__()
        self.dim_to_reduce = dim_to_reduce

    def forward(self, tensor: torch.Tensor):
        return _PoolFn.apply(_btf(tensor, self.dim_to_reduce), ops.POOL_MODES["mean+std"])


class MaxPool1D(torch.nn.Module):
    def __init__(self, dim_to_reduce: int = 2):
        super().__init__()
        self.dim_to_reduce = dim_to_reduce

    def forward(self, tensor: torch.Tensor):
        return _PoolFn.apply(_btf(tensor, self.dim_to_reduce), ops.POOL_MODES["max"])


class IndexPool1D(torch.nn.Module):
    """first / first+cls / last / middle (== last, reference quirk, ref: src/layers/pooling.py:121-122)."""

    def __init__(self, selection_method: str, dim_to_reduce: int):
        super().__init__()
        self.selection_method = selection_method
        self.dim_to_reduce = dim_to_reduce

    def forward(self, tensor: torch.Tensor):
        t = _btf(tensor, self.dim_to_reduce)
        if self.selection_method in ("first", "first+cls", "middle", "last"):
            return _PoolFn.apply(t, ops.POOL_MODES[self.selection_method])
        if self.selection_method == "random":
            idx = random.randint(0, int(t.shape[1]) - 1)
            return torch.clone(t[:, idx, :])
        raise ValueError(f"unknown index {self.selection_method}")


class NoPooling(torch.nn.Module):
    def forward(self, tensor: torch.Tensor):
        return tensor


class QuantilePool1D(torch.nn.Module):
    def __init__(self, dim_to_reduce: int = 2):
        super().__init__()
        raise NotImplementedError("quantile pooling (ref: src/layers/pooling.py:51-67) has no HIP kernel yet "
                                  "(SURVEY 8f row f1)")


class AttentiveStatPool1D(torch.nn.Module):
    def __init__(self, embedding_size: int, dim_to_reduce: int = 2):
        super().__init__()
        raise NotImplementedError("attentive statistics pooling (speechbrain; ref: src/layers/pooling.py:87-106) "
                                  "runs inside the engine (its parameters live in the ParamStore arena): use "
                                  "Wav2vec2FCModule(stat_pooling_type='attentive') or engine.Plan(pooling='attentive')"
                                  " -- w2v2_speaker_amd/asp.py")
