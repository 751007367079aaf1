"""One training step of the reference's hot loop on one GPU of a data-parallel job.

  forward (conv stack -> projection -> SpecAugment mask -> encoder -> pooling -> AAM/CE head)
  -> hand-written backward -> [RCCL all-reduce of gradient buckets on a side HIP stream, overlapped
  with the rest of backward] -> fused Adam with the one-cycle lr / beta1 of this step.

Data parallelism is one process per GPU (``torch.distributed``, backend "nccl" == RCCL over xGMI):
every rank holds a full replica, draws its own minibatch, and the only collective is the SUM
all-reduce of the flat gradient buffer, issued bucket by bucket in the order backward finishes them
(ref: PL ``accelerator: ddp``, config/trainer/trainer.yaml:6-12).  The 1/world of the
gradient mean is folded into the Adam kernel.  LayerDrop-skipped layers contribute zero gradients, so
every rank issues identical collectives regardless of its own LayerDrop draws.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import Plan
from .params import ParamStore
from .spec_augment import compute_mask_indices


class BucketAllReducer:
    """All-reduce contiguous slices of the flat gradient buffer on a side stream as they become final."""

    def __init__(self, store: ParamStore, process_group=None, bucket_merge: int = 2):
        import torch.distributed as dist
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.store = store
        self.ranges, self.members = self.merge_buckets(store.grad_buckets(), bucket_merge)
        self.comm_stream = torch.cuda.Stream() if store.device.type == "cuda" else None
        self.works = []

    @staticmethod
    def merge_buckets(raw: List[Tuple[str, int, int]], bucket_merge: int):
        """Merge neighbouring layer buckets so each collective carries >= ~2 layers (xGMI rings are per-link bound:
        fewer, larger messages; SURVEY 5 "Distributed comm backend").  Returns (ranges: firing bucket -> (start, end),
        members: firing bucket -> raw buckets its collective covers; it fires when the LAST member is final)."""
        ranges, members = {}, {}
        i = 0
        while i < len(raw):
            n, s, e = raw[i]
            j = i
            if n.startswith("layer"):
                while j + 1 < len(raw) and raw[j + 1][0].startswith("layer") and (j - i + 1) < bucket_merge:
                    j += 1
                e = raw[j][2]
            ranges[raw[j][0]] = (s, e)
            members[raw[j][0]] = [raw[k][0] for k in range(i, j + 1)]
            i = j + 1
        return ranges, members

    def bucket_ready(self, name: str) -> None:
        if self.world == 1 or name not in self.ranges:
            return
        s, e = self.ranges[name]
        if e <= s:
            return
        view = self.store.grad[s:e]
        if self.comm_stream is None:       # CPU / gloo test path
            self.works.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            self.works.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def broadcast_parameters(self, root: int = 0, host_counters: Optional[Sequence[int]] = None) -> List[int]:
        """Start-up broadcast (ref: config/trainer/trainer.yaml:6-12 -- PL's DDP wrapper broadcasts the
        module state of rank 0 when it wraps the model): master parameters, optimiser moments and the loss-scale record
        of ``root`` replace every other rank's, then the 16-bit operand copies are rebuilt.  The HOST side of the
        optimiser state travels with them: the store's Adam step counts (bias corrections) and whatever
        ``host_counters`` the caller adds (schedule position, freeze counter) -- returned as ``root`` had them, so that
        a checkpoint loaded on rank 0 only leaves every replica at the same step.  Ranks that have no Adam moments yet
        must not differ from root in that respect (a collective: same tensors on every rank)."""
        extra = [int(c) for c in (host_counters or ())]
        if self.world == 1:
            return extra
        dev = self.store.flat.device
        counts = torch.tensor([len(self.store.replica_state()), len(extra)], device=dev, dtype=torch.int64)
        lo, hi = counts.clone(), counts.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.pg)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            raise RuntimeError("broadcast_parameters: ranks disagree on which state tensors exist (optimiser moments / "
                               "loss scale / host counters); create or load them on every rank first")
        src = self.dist.get_global_rank(self.pg, root) if self.pg is not None else root
        for t in self.store.replica_state():
            self.dist.broadcast(t, src=src, group=self.pg)
        host = torch.tensor([self.store.step_head, self.store.step_body] + extra, device=dev, dtype=torch.int64)
        self.dist.broadcast(host, src=src, group=self.pg)
        host = [int(v) for v in host.tolist()]
        self.store.set_step_counts(host[0], host[1])
        self.store.sync_lowp()
        return host[2:]

    def wait(self) -> None:
        for w in self.works:
            w.wait()                      # makes the current (compute) stream wait for the collective
        self.works = []
        if self.comm_stream is not None and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class SpeakerTrainer:
    def __init__(self, store: ParamStore, plan: Plan, schedule, process_group=None, beta2: float = 0.999,
                 eps: float = 1e-8, layerdrop_seed: int = 1234, mask_seed: int = 7, reducer=None):
        """reducer: an object with bucket_ready(name) / wait() / world (default: BucketAllReducer over
        torch.distributed; comm.CAbiBucketAllReducer runs the collective through the C ABI alone)."""
        assert plan.train
        self.store, self.plan, self.schedule = store, plan, schedule
        self.beta2, self.eps = beta2, eps
        self.step = 0
        self.reducer = reducer if reducer is not None else BucketAllReducer(store, process_group)
        self.world = self.reducer.world
        self._ld_rng = np.random.RandomState(layerdrop_seed)
        self._mask_rng = np.random.RandomState(mask_seed)

    def broadcast_state(self, root: int = 0, host_counters: Optional[Sequence[int]] = None) -> List[int]:
        """Every replica takes ``root``'s parameters, moments, loss scale AND step position (this trainer's ``step`` =
        the lr / beta1 schedule index, the store's Adam step counts, plus the caller's ``host_counters``, returned as
        root had them).  What a rank-0-only checkpoint load needs before the first step."""
        got = self.reducer.broadcast_parameters(root, [self.step] + [int(c) for c in (host_counters or ())])
        self.step = got[0]
        return got[1:]

    def sample_layerdrop(self) -> Tuple[int, ...]:
        """HF:698-709: each layer is skipped with probability ``layerdrop`` (host RNG)."""
        p = self.plan.reg.layerdrop
        if p <= 0:
            return ()
        u = self._ld_rng.rand(self.plan.cfg.num_hidden_layers)
        return tuple(int(i) for i in np.nonzero(u < p)[0])

    def sample_time_mask(self) -> Optional[torch.Tensor]:
        reg, plan = self.plan.reg, self.plan
        if reg.mask_time_prob <= 0 or plan.cls or plan.paired:
            return None
        m = compute_mask_indices((plan.B, plan.T0), reg.mask_time_prob, reg.mask_time_length,
                                 plan.cfg.mask_time_min_masks, rng=self._mask_rng)
        return torch.from_numpy(m.astype(np.uint8)).to(plan.dev, non_blocking=True)

    def sample_feature_mask(self) -> Optional[torch.Tensor]:
        """HF:1294-1304: [B, hidden_size] mask of feature channels, drawn AFTER the time mask from the same stream."""
        reg, plan = self.plan.reg, self.plan
        if reg.mask_feature_prob <= 0 or plan.cls or plan.paired:
            return None
        m = compute_mask_indices((plan.B, plan.cfg.hidden_size), reg.mask_feature_prob, reg.mask_feature_length,
                                 getattr(plan.cfg, "mask_feature_min_masks", 0), rng=self._mask_rng)
        return torch.from_numpy(m.astype(np.uint8)).to(plan.dev, non_blocking=True)

    def train_step_frozen_encoder(self, frozen_plan: Plan, wav: torch.Tensor, label: torch.Tensor):
        """Step while the whole wav2vec2 network is frozen (ref: wav2vec2_fc.py:339-347 + PL ``freeze()`` =
        requires_grad False AND eval mode): eval-mode forward, head forward/backward, Adam on the head only."""
        store = self.store
        store.zero_grad()
        frozen_plan.embed(wav, None, (), self.step)
        loss, softmax = frozen_plan.head_forward_backward(label)
        self.reducer.bucket_ready("head")
        self.reducer.wait()
        lr, beta1 = self.schedule.at(self.step)
        store.adam_step(lr, beta1, self.beta2, self.eps, grad_scale=1.0 / self.world, head_only=True)
        self.step += 1
        return loss, softmax

    def train_step(self, wav: torch.Tensor, label: torch.Tensor, mask: Optional[torch.Tensor] = None,
                   skip_layers: Optional[Sequence[int]] = None, feature_mask: Optional[torch.Tensor] = None):
        """ref: speaker_recognition_module.py:207-220 (_train_step_ce_loss) + PL backward/optimizer step.
        Returns (loss, softmax) as device tensors; no host sync."""
        plan, store = self.plan, self.store
        if skip_layers is None:
            skip_layers = self.sample_layerdrop()
        if mask is None:
            mask = self.sample_time_mask()
        if feature_mask is None:
            feature_mask = self.sample_feature_mask()
        store.zero_grad(tuple(skip_layers) if plan.grouped else None)
        plan.embed(wav, mask, skip_layers, self.step, feature_mask)
        loss, softmax = plan.head_forward_backward(label)
        plan.backward(on_bucket_ready=self.reducer.bucket_ready)
        self.reducer.wait()
        lr, beta1 = self.schedule.at(self.step)
        store.adam_step(lr, beta1, self.beta2, self.eps, grad_scale=1.0 / self.world)
        self.step += 1
        return loss, softmax
