"""Plain Euler-Maruyama simulator of the *model* SDE (reference: core/euler_maruyama.py:11-45).
Used by the parameter pre-training stage; the user's drift/diffusion are Python callables."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Optional

import torch
from torch import Tensor

from .sde import SDE


def euler_maruyama(sde: SDE, x0: Tensor, theta: Tensor, time_horizon: float, dt: float,
                   positive_dims: Sequence[int] = (), noise: Optional[Tensor] = None) -> Tensor:
    """Returns the trajectory ``[batch, n_steps+1, state_dim]``; positive dims are clamped at 1e-6."""
    if dt <= 0:
        raise ValueError(f"dt must be positive, got {dt}")
    if time_horizon <= 0:
        raise ValueError(f"time_horizon must be positive, got {time_horizon}")
    n_steps = round(time_horizon / dt)
    batch, state_dim = x0.shape
    if noise is None:
        noise = torch.randn(batch, n_steps, state_dim, device=x0.device, dtype=x0.dtype)
    pos = list(positive_dims)
    root_dt = dt ** 0.5
    states = [x0]
    x = x0
    for k in range(n_steps):
        shock = torch.einsum("bij,bj->bi", sde.diffusion(x, theta), noise[:, k])
        x = x + sde.drift(x, theta) * dt + shock * root_dt
        if pos:
            floor = torch.full_like(x, -float("inf"))
            floor[:, pos] = 1e-6
            x = torch.maximum(x, floor)
        states.append(x)
    return torch.stack(states, dim=1)
