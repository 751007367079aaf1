"""Mirror of ref: src/optim/loss/cross_entropy.py -- CE + softmax prediction on logits, via the HIP row
kernel (``w2v2_aam_softmax_fwd_bwd`` with margin < 0)."""
from __future__ import annotations

import torch

from ... import ops


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label):
        B, C = logits.shape
        ldc = (C + 7) // 8 * 8
        lg = torch.zeros(B, ldc, dtype=torch.float32, device=logits.device)
        lg[:, :C] = logits
        sm = torch.zeros_like(lg)
        dl = torch.zeros_like(lg)
        rows = torch.empty(B, dtype=torch.float32, device=logits.device)
        ops.aam_softmax_fwd_bwd(lg, label, sm, rows, dl, None, None, None, None, None, B, C, ldc, -1.0, 1.0)
        ctx.save_for_backward(dl[:, :C])
        out = sm[:, :C].clone()
        ctx.mark_non_differentiable(out)
        return rows.mean(), out

    @staticmethod
    def backward(ctx, dloss, _):
        (dl,) = ctx.saved_tensors
        return dl * dloss, None


class CrossEntropyLoss(torch.nn.Module):
    def forward(self, logits: torch.Tensor, label_indexes: torch.Tensor):
        return _CEFn.apply(logits.float().contiguous(), label_indexes)
