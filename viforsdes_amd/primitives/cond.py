"""adaLN-zero style conditioning (reference: primitives/cond.py:12-72).

``CondModulator`` maps a conditioning vector to ``3 * branches`` chunks (scale, shift, gate per
branch).  Its linear is zero-initialised, so a fresh block is the identity map."""
from __future__ import annotations

from dataclasses import dataclass

from torch import Tensor, nn

from .initializer import zero_linear_


@dataclass
class CondBranch:
    scale: Tensor
    shift: Tensor
    gate_value: Tensor

    def affine(self, x: Tensor) -> Tensor:
        return x * (1 + self.scale) + self.shift

    def gate(self, x: Tensor) -> Tensor:
        return x * self.gate_value


class CondModulator(nn.Module):
    def __init__(self, cond_dim: int, hidden_dim: int, *, branches: int = 1) -> None:
        super().__init__()
        if branches <= 0:
            raise ValueError("branches must be positive")
        self.branch_count = branches
        self.net = nn.Sequential(nn.SiLU(), zero_linear_(nn.Linear(cond_dim, 3 * branches * hidden_dim)))

    def forward(self, *, cond: Tensor) -> tuple[CondBranch, ...]:
        pieces = self.net(cond).chunk(3 * self.branch_count, dim=-1)
        return tuple(CondBranch(*pieces[3 * i:3 * i + 3]) for i in range(self.branch_count))
