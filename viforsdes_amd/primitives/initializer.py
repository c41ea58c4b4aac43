"""Weight initialisation policy: truncated normal (std 0.02) for transformer linears, zeros for
modulators / gates (reference: primitives/initializer.py:10-45)."""
from __future__ import annotations

from torch import nn

TRUNC_STD = 0.02


def init_linear_(layer: nn.Linear, std: float = TRUNC_STD) -> nn.Linear:
    nn.init.trunc_normal_(layer.weight, mean=0.0, std=std)
    if layer.bias is not None:
        nn.init.zeros_(layer.bias)
    return layer


def zero_linear_(layer: nn.Linear) -> nn.Linear:
    nn.init.zeros_(layer.weight)
    if layer.bias is not None:
        nn.init.zeros_(layer.bias)
    return layer
