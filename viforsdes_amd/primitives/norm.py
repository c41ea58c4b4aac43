"""Normalisation layers (reference: primitives/norm.py:10-34)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor, nn


class RMS(nn.Module):
    """RMS norm evaluated in fp32 whatever the input dtype; ``weight`` may be frozen."""

    def __init__(self, dim: int, eps: float = 1e-6, requires_grad: bool = True) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=requires_grad)

    def forward(self, x: Tensor) -> Tensor:
        xf = x.float()
        inv = torch.rsqrt(xf.square().mean(dim=-1, keepdim=True) + self.eps)
        return (xf * inv * self.weight.float()).to(x.dtype)


@dataclass(frozen=True)
class LayerNormConfig:
    eps: float = 1e-5
    affine: bool = True

    def build(self, *, dim: int) -> nn.Module:
        return nn.LayerNorm(dim, eps=self.eps, elementwise_affine=self.affine)
