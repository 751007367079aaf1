"""Attribute handles that stand in for the HuggingFace sub-modules the reference's callers touch:
``wav2vec.model.feature_extractor.requires_grad_(False)`` and ``wav2vec.freeze()`` / ``unfreeze()``
(ref: src/lightning_modules/speaker/wav2vec2_fc.py:339-361), ``wav2vec.model.feature_projection`` /
``.encoder`` / ``.num_features`` (ref: wav2vec2_paired_input.py:171-200, wav2vec2_fc.py:247,450).

The parameters live in the flat ParamStore arena, not in nn.Modules; a handle maps the reference's attribute
path to the arena prefix and turns ``requires_grad_`` into the store's trainability switches."""
from __future__ import annotations

from typing import Callable, Iterator, Optional

import torch

from ..params import ParamStore, W2V_PREFIX


class SubmoduleHandle:
    def __init__(self, store: ParamStore, prefix: str, on_requires_grad: Optional[Callable[[bool], None]] = None):
        self.store, self.prefix, self._on = store, W2V_PREFIX + prefix, on_requires_grad
        self._requires_grad = True

    def named_parameters(self) -> Iterator:
        for n in self.store.shapes:
            if n.startswith(self.prefix):
                yield n[len(self.prefix):], self.store.p(n)

    def parameters(self) -> Iterator[torch.Tensor]:
        for _, p in self.named_parameters():
            yield p

    def state_dict(self):
        return {n: p.detach().clone().cpu() for n, p in self.named_parameters()}

    def requires_grad_(self, requires_grad: bool = True) -> "SubmoduleHandle":
        if self._on is not None:
            self._on(bool(requires_grad))
        self._requires_grad = bool(requires_grad)
        return self


class ModelHandle:
    """``wrapper.model`` of the reference (the HF Wav2Vec2Model): the three sub-modules its callers reach for."""

    def __init__(self, store: ParamStore, on_body_grad: Optional[Callable[[bool], None]] = None):
        def cnn(flag: bool) -> None:
            # ref: wav2vec2_fc.py:347,361 ``feature_extractor.requires_grad_(False)`` (the reference's default)
            if flag and store.freeze_cnn:
                raise RuntimeError("this ParamStore was built with freeze_cnn=True (no gradient buffers for the conv "
                                   "feature extractor): construct it with completely_freeze_feature_extractor=False")
            store.cnn_runtime_frozen = not flag
        self.feature_extractor = SubmoduleHandle(store, "feature_extractor.", cnn)
        self.feature_projection = SubmoduleHandle(store, "feature_projection.", on_body_grad)
        self.encoder = SubmoduleHandle(store, "encoder.", on_body_grad)
        self.config = store.cfg
