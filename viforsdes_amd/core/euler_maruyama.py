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
    floor = _floor_vector(pos, state_dim, x0.device, x0.dtype) if pos else None
    for k in range(n_steps):
        shock = torch.einsum("bij,bj->bi", sde.diffusion(x, theta), noise[:, k])
        x = x + sde.drift(x, theta) * dt + shock * root_dt
        if floor is not None:
            x = torch.maximum(x, floor)  # 1e-6 on the positive dims, -inf elsewhere
        states.append(x)
    return torch.stack(states, dim=1)


_FLOORS: dict = {}


def _floor_vector(pos: list[int], state_dim: int, device: torch.device, dtype: torch.dtype) -> Tensor:
    """[state_dim] clamp floor, built once per (dims, device): building it inside the time loop costs three kernels per Euler
    step and an index upload that cannot be captured into a HIP graph."""
    key = (tuple(pos), state_dim, str(device), dtype)
    f = _FLOORS.get(key)
    if f is None:
        host = torch.full((state_dim,), -float("inf"), dtype=dtype)
        host[pos] = 1e-6
        f = _FLOORS[key] = host.to(device)
    return f
