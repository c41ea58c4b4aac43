"""``lower_bound``: max(x, bound) whose gradient also flows when it would push x back up
(reference: primitives/bounds.py:10-31).  The fused kernels implement the same rule for the
Cholesky diagonal (csrc/vsde_head.hip, backward, "bounds.py:20")."""
from __future__ import annotations

import torch
from torch import Tensor


class _LowerBound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, bound: Tensor) -> Tensor:
        ctx.save_for_backward(x, bound)
        return torch.maximum(x, bound)

    @staticmethod
    def backward(ctx, g: Tensor):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return g * keep, None


def lower_bound(x: Tensor, bound: float | Tensor) -> Tensor:
    return _LowerBound.apply(x, torch.as_tensor(bound, dtype=x.dtype, device=x.device))
