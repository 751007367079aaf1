"""Learning-rate / momentum schedules computed on the host (scalars fed to the fused Adam kernel).

``OneCycle`` = torch ``OneCycleLR`` with the reference's settings (ref: config/optim/schedule/one_cycle.yaml:3-20,
wired at src/main.py:323-335): cosine anneal, two phases, pct_start 0.3, div_factor 25, final_div 1e4 and
-- torch default ``cycle_momentum=True`` -- Adam's beta1 cycled 0.95 -> 0.85 -> 0.95 (quirk Q11)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple


@dataclass
class OneCycle:
    max_lr: float
    total_steps: int
    pct_start: float = 0.3
    div_factor: float = 25.0
    final_div_factor: float = 1e4
    base_momentum: float = 0.85
    max_momentum: float = 0.95

    def at(self, step: int) -> Tuple[float, float]:
        """(lr, beta1) used by optimiser step number ``step`` (0-based)."""
        if step >= self.total_steps:
            raise ValueError(f"Tried to step {step + 1} times. The specified number of total steps is {self.total_steps}")
        initial_lr = self.max_lr / self.div_factor
        min_lr = initial_lr / self.final_div_factor
        end1 = float(self.pct_start * self.total_steps) - 1
        end2 = self.total_steps - 1

        def cos(a, b, pct):
            return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

        if step <= end1:
            pct = step / end1
            return cos(initial_lr, self.max_lr, pct), cos(self.max_momentum, self.base_momentum, pct)
        pct = (step - end1) / (end2 - end1)
        return cos(self.max_lr, min_lr, pct), cos(self.base_momentum, self.max_momentum, pct)


@dataclass
class Constant:
    lr: float
    beta1: float = 0.9

    def at(self, step: int) -> Tuple[float, float]:
        return self.lr, self.beta1
