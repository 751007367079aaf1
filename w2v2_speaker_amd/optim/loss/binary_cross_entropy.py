"""Mirror of ref: src/optim/loss/binary_cross_entropy.py:16-40 -- BCE-with-logits (mean over the batch) + sigmoid
prediction, on the HIP head kernel (``w2v2_bce_head_fwd_bwd`` with a unit 1 x 1 "linear layer": logit = 1 * x + 0)."""
from __future__ import annotations

import torch

from ... import ops


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label):
        x = logits.reshape(-1, 1).float().contiguous()
        B, dev = x.shape[0], x.device
        one, zero = torch.ones(1, device=dev), torch.zeros(1, device=dev)
        prob, rows = torch.empty(B, device=dev), torch.empty(B, device=dev)
        dl, de = torch.empty(B, device=dev), torch.empty(B, 1, device=dev)
        dw, db = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        ops.bce_head_fwd_bwd(x, one, zero, label.reshape(-1).to(torch.int64), prob, rows, dl, de, dw, db, B, 1)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(prob)
        loss = torch.empty((), device=dev)
        ops.mean(rows, loss)
        return loss, prob

    @staticmethod
    def backward(ctx, dloss, _):
        (dl,) = ctx.saved_tensors
        return (dl * dloss).reshape(ctx.shape), None


class BinaryCrossEntropyLoss(torch.nn.Module):
    def forward(self, logits: torch.Tensor, label_indexes: torch.Tensor):
        """logits [B, 1] (or [B]) on the GPU, labels in {0, 1} -> (loss, prediction = sigmoid(logits) [B])."""
        return _BCEFn.apply(logits, label_indexes)
