"""Euler-Maruyama simulator of the *model* SDE (reference: core/euler_maruyama.py:11-45), used by the parameter
pre-training stage.

SDEs whose drift / diffusion are built into the HIP library (``sde.builtin_kind`` in ``_hip.SDE_KINDS``: the example
Ornstein-Uhlenbeck and Lotka-Volterra models and the benchmark's linear-diagonal one) run as ONE kernel forward and ONE
backward (csrc/vsde_sde.hip: a thread per path, the time loop in the kernel) when the tensors live on the GPU; any other
SDE -- the user's drift / diffusion are Python callables -- takes the torch loop below, which is also the specification."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Optional

import torch
from torch import Tensor

from .sde import SDE, builtin_sde_kind


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
    kind = builtin_sde_kind(sde)
    if kind is not None and x0.is_cuda and x0.dtype == torch.float32 and HIP_SIMULATOR:
        from .. import _hip
        if kind in _hip.SDE_KINDS and noise.shape == (batch, n_steps, state_dim):
            return _BuiltinEulerMaruyama.apply(x0, theta, noise, kind, float(dt), tuple(pos))
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


HIP_SIMULATOR = True  # set False to force the torch loop (A/B tests)


class _BuiltinEulerMaruyama(torch.autograd.Function):
    """Trajectory of a built-in SDE through the HIP simulator; differentiable in theta and x0 (noise is a constant)."""

    @staticmethod
    def forward(ctx, x0, theta, noise, kind, dt, pos):
        from .. import _hip
        theta_c, noise_c = theta.detach().float().contiguous(), noise.detach().float().contiguous()
        traj = _hip.euler_maruyama_fwd(kind, x0.detach(), theta_c, noise_c, dt, pos)
        ctx.save_for_backward(theta_c, noise_c, traj)
        ctx.meta = (kind, dt, pos, theta.dtype)
        return traj

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_traj):
        from .. import _hip
        theta, noise, traj = ctx.saved_tensors
        kind, dt, pos, tdtype = ctx.meta
        g_x0, g_theta = _hip.euler_maruyama_bwd(kind, theta, noise, traj, g_traj.contiguous(), dt, pos)
        return g_x0, g_theta.to(tdtype), None, None, None, None


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
