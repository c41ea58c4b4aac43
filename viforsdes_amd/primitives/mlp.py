"""SwiGLU feed-forward (reference: primitives/mlp.py:11-54): silu(a) * b with [a | b] = W_in x."""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import fused
from .initializer import init_linear_


class SwiGLU(nn.Module):
    def __init__(self, in_dim: int, hidden_dim: int, *, bias: bool = True) -> None:
        super().__init__()
        self.in_dim, self.hidden_dim = in_dim, hidden_dim
        self.input_proj = init_linear_(nn.Linear(in_dim, 2 * hidden_dim, bias=bias))
        self.output_proj = init_linear_(nn.Linear(hidden_dim, in_dim, bias=bias))

    def forward(self, x: Tensor) -> Tensor:
        u = self.input_proj(x)
        if fused.ENABLED and u.is_cuda and u.dtype in (torch.float32, torch.bfloat16):
            return self.output_proj(fused.swiglu(u))
        a, b = u.chunk(2, dim=-1)
        return self.output_proj(F.silu(a) * b)
