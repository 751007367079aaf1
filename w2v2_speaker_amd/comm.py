"""Gradient all-reduce through the C ABI alone (include/w2v2_hip.h "collective": w2v2_comm_* over librccl.so).

ref: config/trainer/trainer.yaml:6-12 (PL ``accelerator: ddp``).  ``trainer.BucketAllReducer`` (torch.distributed "nccl"
== RCCL) stays the default binding; ``CAbiBucketAllReducer`` is the same schedule -- contiguous gradient buckets,
reduced on a side HIP stream as soon as backward finalises them -- for a host that has no torch.distributed process
group: the 128-byte RCCL id travels through a ``torch.distributed.TCPStore`` (a plain key-value socket, no process
group), a file, or whatever the launcher provides."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib


class RcclComm:
    """One RCCL communicator owned through the C ABI."""

    def __init__(self, rank: int, world: int, device: int, unique_id: bytes):
        assert len(unique_id) == 128
        self.rank, self.world, self.device = rank, world, device
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        _lib.check(_lib.load().w2v2_comm_init(C.byref(self._h), buf, rank, world, device), "comm_init")

    @classmethod
    def loopback(cls, world: int, device: int = 0) -> "RcclComm":
        """w2v2_comm_init_loopback: rank 0 of a ``world``-rank job whose peers hold identical buffers (SUM all-reduce =
        x world, broadcast = identity).  Runs the world > 1 path of a reducer on a one-GPU box; no RCCL involved."""
        self = cls.__new__(cls)
        self.rank, self.world, self.device = 0, world, device
        self._h = C.c_void_p()
        _lib.check(_lib.load().w2v2_comm_init_loopback(C.byref(self._h), world, device), "comm_init_loopback")
        return self

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().w2v2_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    @classmethod
    def from_store(cls, rank: int, world: int, device: int, host: str = "127.0.0.1", port: int = 29555,
                   timeout_s: float = 300.0) -> "RcclComm":
        """Rendezvous over a TCPStore on (host, port): rank 0 creates the id, the others read it."""
        from datetime import timedelta
        from torch.distributed import TCPStore
        store = TCPStore(host, port, world, is_master=(rank == 0), timeout=timedelta(seconds=timeout_s))
        if rank == 0:
            uid = cls.unique_id()
            store.set("w2v2_rccl_id", uid)
        else:
            uid = bytes(store.get("w2v2_rccl_id"))
        comm = cls(rank, world, device, uid)
        comm._store = store            # keep the server alive until every rank has initialised
        return comm

    def all_reduce_(self, t: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        """SUM all-reduce of a contiguous f32 CUDA tensor, in place, enqueued on ``stream`` (default: current)."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        s = (stream or torch.cuda.current_stream()).cuda_stream
        _lib.check(_lib.load().w2v2_allreduce_async(self._h, t.data_ptr(), t.numel(), s), "allreduce_async")

    def broadcast_(self, t: torch.Tensor, root: int = 0, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Broadcast a contiguous CUDA tensor (any dtype: bytes) from ``root``, in place, enqueued on ``stream``."""
        assert t.is_cuda and t.is_contiguous()
        s = (stream or torch.cuda.current_stream()).cuda_stream
        _lib.check(_lib.load().w2v2_broadcast_async(self._h, t.data_ptr(), t.numel() * t.element_size(), root, s),
                   "broadcast_async")

    def destroy(self) -> None:
        if self._h:
            _lib.check(_lib.load().w2v2_comm_destroy(self._h), "comm_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        import sys
        if sys is None or sys.is_finalizing():       # interpreter teardown: RCCL may already be unloaded
            return
        try:
            self.destroy()
        except Exception:
            pass


def signature_limbs(values) -> list:
    """Rank-agreement signature for an f32 SUM all-reduce: every non-negative integer in ``values`` is cut into six
    8-bit limbs x (covers 2^48), and the vector carries x and x^2 for each.  With x <= 255 every summed quantity stays
    <= world * 65025 < 2^24 for world <= 258, i.e. EXACT in f32 -- both on the wire and in whatever order the ranks
    add (13-bit limbs squared to 6.7e7 and rounded, so agreeing ranks failed the check ~25-39 % of the time,
    wav2vec2-base at start-up among them)."""
    limbs = []
    for v in values:
        v = int(v)
        if v < 0 or v >= 1 << 48:
            raise ValueError(f"signature value out of range: {v}")
        limbs += [(v >> (8 * i)) & 0xFF for i in range(6)]
    return [float(x) for x in limbs] + [float(x * x) for x in limbs]


def signature_agrees(summed, world: int) -> bool:
    """``summed`` = signature_limbs(...) summed over ``world`` ranks.  All ranks held the same x iff
    world * sum(x^2) == (sum x)^2 (Cauchy-Schwarz with equality); integer arithmetic on exactly represented values."""
    if world > 258:
        raise ValueError("signature check is exact in f32 for at most 258 ranks")
    h = len(summed) // 2
    return all(world * int(summed[h + i]) == int(summed[i]) ** 2 for i in range(h))


class CAbiBucketAllReducer:
    """Drop-in for trainer.BucketAllReducer (same ``bucket_ready`` / ``wait`` / ``world`` / ``ranges`` / ``members``)
    whose collective is w2v2_allreduce_async on a side HIP stream."""

    def __init__(self, store, comm: RcclComm, bucket_merge: int = 2):
        from .trainer import BucketAllReducer
        self.comm, self.store, self.world = comm, store, comm.world
        self.ranges, self.members = BucketAllReducer.merge_buckets(store.grad_buckets(), bucket_merge)
        self.comm_stream = torch.cuda.Stream()
        self._issued = False

    def bucket_ready(self, name: str) -> None:
        if self.world == 1 or name not in self.ranges:
            return
        s, e = self.ranges[name]
        if e <= s:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        self.comm.all_reduce_(self.store.grad[s:e], self.comm_stream)
        self._issued = True

    def broadcast_parameters(self, root: int = 0, host_counters=None):
        """Same contract as trainer.BucketAllReducer.broadcast_parameters (device state + the host-side step counters),
        through w2v2_broadcast_async.  Everything is enqueued on the communicator's own stream (the one the gradient
        all-reduces use), ordered against the compute stream on both sides; before any broadcast the ranks check,
        through one tiny all-reduce, that they agree on how many state tensors and bytes there are -- mismatched
        ncclBroadcast sequences would hang instead of failing."""
        extra = [int(c) for c in (host_counters or ())]
        if self.world == 1:
            return extra
        ts = self.store.replica_state()
        dev = self.store.flat.device
        cur = torch.cuda.current_stream()
        nbytes = sum(t.numel() * t.element_size() for t in ts)
        # both small tensors are created (H2D copies on the CURRENT stream) before the side stream is ordered behind it
        chk = torch.tensor(signature_limbs([len(ts), len(extra), nbytes]), device=dev, dtype=torch.float32)
        host = torch.tensor([self.store.step_head, self.store.step_body] + extra, device=dev, dtype=torch.int64)
        chk.record_stream(self.comm_stream)
        host.record_stream(self.comm_stream)
        self.comm_stream.wait_stream(cur)
        self.comm.all_reduce_(chk, self.comm_stream)
        self.comm_stream.synchronize()
        if not signature_agrees(chk.tolist(), self.comm.world):
            raise RuntimeError("broadcast_parameters: ranks disagree on which state tensors exist (optimiser moments / "
                               "loss scale / host counters); create or load them on every rank first")
        for t in ts:
            self.comm.broadcast_(t, root, self.comm_stream)
        self.comm.broadcast_(host, root, self.comm_stream)
        cur.wait_stream(self.comm_stream)
        self.comm_stream.synchronize()
        host = [int(v) for v in host.tolist()]
        self.store.set_step_counts(host[0], host[1])
        self.store.sync_lowp()
        return host[2:]

    def wait(self) -> None:
        if self._issued:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._issued = False
