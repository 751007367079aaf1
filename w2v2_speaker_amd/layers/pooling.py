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
        out = torch.empty(B, ops.POOL_WIDTH.get(mode, 1) * H, dtype=torch.float32, device=x.device)
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
    """ref: src/layers/pooling.py:51-67 -- the 0 / .25 / .5 / .75 / 1 quantiles over time, stacked [B, 5 F]."""

    def __init__(self, dim_to_reduce: int = 2):
        super().__init__()
        self.dim_to_reduce = dim_to_reduce
        self.quantiles = torch.tensor([0, 0.25, 0.5, 0.75, 1]).detach()

    def forward(self, tensor: torch.Tensor):
        return _PoolFn.apply(_btf(tensor, self.dim_to_reduce), ops.POOL_MODES["quantile"])


class _AspStore:
    """The slice of ParamStore's interface that asp.AttentivePool needs, over the parameters of ONE pooling module."""

    def __init__(self, channels: int, attention_channels: int, device, act_dtype: torch.dtype):
        from ..asp import ASP_PREFIX, asp_param_shapes
        self.prefix = ASP_PREFIX
        self.shapes = asp_param_shapes(channels, attention_channels)
        self.device, self.act_dtype = torch.device(device), act_dtype
        self.offsets, off = {}, 0
        for n, shp in self.shapes.items():
            self.offsets[n] = off
            off += (int(torch.tensor(shp).prod()) + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_lp = torch.zeros(off, dtype=act_dtype, device=device) if ops.is16(act_dtype) else None
        self.asp_running = torch.cat([torch.zeros(attention_channels), torch.ones(attention_channels)]).to(device)
        self.asp_batches_tracked = 0

    def _view(self, buf, name):
        shp, o = self.shapes[name], self.offsets[name]
        n = 1
        for d in shp:
            n *= d
        return buf[o:o + n].view(*shp)

    def p(self, name): return self._view(self.flat, name)
    def g(self, name): return self._view(self.grad, name)
    def w(self, name): return self._view(self.flat_lp if self.flat_lp is not None else self.flat, name)

    def sync_lowp(self):
        if self.flat_lp is not None:
            ops.cast(self.flat, self.flat_lp)


class _AspFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, module):
        B, T, C = x.shape
        pool = module._pool(B, T, x.requires_grad or flat.requires_grad or module.training)
        pool.x.copy_(x.reshape(B * T, C))
        module._store.sync_lowp()
        out = pool.forward().clone()
        ctx.pool, ctx.module, ctx.shape = pool, module, (B, T, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        pool, st = ctx.pool, ctx.module._store
        if not pool.train:
            raise RuntimeError("backward through AttentiveStatPool1D needs module.train()")
        st.grad.zero_()
        pool.dx.zero_()
        pool.backward(dout.contiguous().float())
        return pool.dx.view(*ctx.shape).to(dout.dtype).clone(), st.grad.clone(), None


class AttentiveStatPool1D(torch.nn.Module):
    """ref: src/layers/pooling.py:87-106 -> speechbrain ``AttentiveStatisticsPooling(embedding_size)`` (attention
    channels 128, global context): [B, T, F] (dim_to_reduce=1) or [B, F, T] (dim_to_reduce=2) -> [B, 2 F] =
    cat(weighted mean, weighted std).  The arithmetic is asp.AttentivePool (csrc/asp.hip + the GEMM kernels); the six
    parameter tensors are ONE flat ``nn.Parameter`` whose named views carry the speechbrain state-dict names
    (``pooling_layer.tdnn.conv.conv.weight`` ...), BatchNorm running statistics are buffers.  speechbrain is not
    available here: the definition is the published one restated in oracle.attentive_stat_pool (parity unpinned)."""

    def __init__(self, embedding_size: int, dim_to_reduce: int = 2, *, device="cuda",
                 act_dtype: torch.dtype = torch.float32, attention_channels: int = 128, init_seed: int = 0):
        super().__init__()
        self.dim_to_reduce, self.embedding_size = dim_to_reduce, embedding_size
        self._store = _AspStore(embedding_size, attention_channels, device, act_dtype)
        st = self._store
        g = torch.Generator().manual_seed(init_seed)
        for n, shp in st.shapes.items():
            if n.endswith("norm.norm.weight"):
                t = torch.ones(shp)
            elif n.endswith("bias"):
                t = torch.zeros(shp)
            else:                                    # torch Conv1d default: kaiming-uniform, bound 1/sqrt(fan_in)
                t = (torch.rand(shp, generator=g) * 2 - 1) / (shp[1] * shp[2]) ** 0.5
            st.p(n).copy_(t.to(st.device))
        self.flat = torch.nn.Parameter(st.flat)          # the module's parameters (one arena)
        self._pools = {}

    def named_views(self):
        """speechbrain-named views of the flat parameter (``pooling_layer.*``) and the BatchNorm buffers."""
        pre = len("stat_pooling.")
        A = self._store.asp_running.numel() // 2
        out = {n[pre:]: self._store.p(n) for n in self._store.shapes}
        out["pooling_layer.tdnn.norm.norm.running_mean"] = self._store.asp_running[:A]
        out["pooling_layer.tdnn.norm.norm.running_var"] = self._store.asp_running[A:]
        return out

    def _pool(self, B: int, T: int, train: bool):
        from ..asp import AttentivePool
        key = (B, T, train)
        if key not in self._pools:
            if len(self._pools) >= 4:
                self._pools.pop(next(iter(self._pools)))
            st = self._store
            C = self.embedding_size
            full = torch.zeros((B * T + 63) // 64 * 64, C, dtype=st.act_dtype, device=st.device)
            x = full[:B * T]
            x._w2v2_padded = full
            emb = torch.empty(B, 2 * C, dtype=torch.float32, device=st.device)
            dx = torch.zeros(B * T, C, dtype=st.act_dtype, device=st.device) if train else None
            self._pools[key] = AttentivePool(st, x, emb, dx, B, T, train)
        return self._pools[key]

    def forward(self, tensor: torch.Tensor):
        t = _btf(tensor, self.dim_to_reduce).contiguous().to(self._store.device)
        pooled = _AspFn.apply(t, self.flat, self)
        return pooled if pooled.dim() == 2 else pooled[None, :]
