"""Mirror of ref: src/optim/loss/aam_softmax.py -- ``AngularAdditiveMarginSoftMaxLoss`` on the HIP head
kernels (heads.ClassifierHead): same ctor, same attributes (``margin``, ``scale``, ``fc_weights``), same
``forward(x, label) -> (loss, prediction)``."""
from __future__ import annotations

import math

import torch

from ...heads import ClassifierHead


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, label, module):
        head = module._head(x.shape[0], x.device, any(ctx.needs_input_grad))
        head.emb.copy_(x)
        loss, sm = head.forward_backward(label)
        ctx.head = head
        sm = sm.clone()
        ctx.mark_non_differentiable(sm)
        return loss.clone(), sm

    @staticmethod
    def backward(ctx, dloss, _dsm):
        head = ctx.head
        if not head.train:
            raise RuntimeError("head was run without grad enabled")
        return head.demb * dloss, head.w_grad * dloss, None, None


class AngularAdditiveMarginSoftMaxLoss(torch.nn.Module):
    def __init__(self, input_features, output_features, margin=0.3, scale=15, easy_margin=False, *,
                 device="cuda", act_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.margin, self.scale, self.input_features = margin, scale, input_features
        self.easy_margin = easy_margin
        w = torch.empty(output_features, input_features, device=device)
        torch.nn.init.xavier_normal_(w, gain=1)                 # ref: aam_softmax.py:38
        self.fc_weights = torch.nn.Parameter(w, requires_grad=True)
        self.cos_m, self.sin_m = math.cos(margin), math.sin(margin)
        self.th = math.cos(math.pi - margin)
        self.mm = math.sin(math.pi - margin) * margin
        self.act_dtype = act_dtype
        self._heads = {}

    def _head(self, batch, device, train) -> ClassifierHead:
        key = (batch, train, self.fc_weights.data_ptr())
        if key not in self._heads:
            C, E = self.fc_weights.shape
            wlp = (torch.empty(C, E, dtype=self.act_dtype, device=device)
                   if self.act_dtype != torch.float32 else self.fc_weights.data)
            emb = torch.empty(batch, E, dtype=torch.float32, device=device)
            head = ClassifierHead("aam", batch, E, C, w_master=self.fc_weights.data, w_operand=wlp,
                                  w_grad=torch.zeros_like(self.fc_weights.data), emb=emb,
                                  act_dtype=self.act_dtype, train=train, margin=self.margin, scale=self.scale,
                                  easy_margin=self.easy_margin)
            self._heads[key] = (head, wlp)
        head, wlp = self._heads[key]
        if wlp is not self.fc_weights.data:
            from ... import ops
            ops.cast(self.fc_weights.data, wlp)          # the parameter may have been stepped by any optimiser
        return head

    def forward(self, x, label=None):
        assert x.size()[0] == label.size()[0]
        assert x.size()[1] == self.input_features
        return _HeadFn.apply(x.float(), self.fc_weights, label, self)
