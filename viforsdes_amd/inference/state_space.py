"""Latent <-> constrained state map (reference: inference/state_space.py:8-38): softplus on the
positive dimensions, identity elsewhere."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn import functional as F


class StateSpace:
    def __init__(self, dim: int, positive_dims: list[int] | None = None) -> None:
        pos = list(positive_dims or [])
        if dim < 1:
            raise ValueError(f"dim must be >= 1, got {dim}")
        if any(d < 0 or d >= dim for d in pos):
            raise ValueError(f"positive_dims must be in [0, {dim}), got {pos}")
        if len(set(pos)) != len(pos):
            raise ValueError(f"positive_dims must be unique, got {pos}")
        self.dim, self.positive_dims = dim, pos
        self._masks: dict = {}

    def _mask(self, like: Tensor) -> Tensor:
        """Boolean [dim] mask of the positive dims on ``like``'s device (cached: building it copies from the
        host, which must not happen inside a captured HIP graph)."""
        m = self._masks.get(like.device)
        if m is None:
            host = torch.zeros(self.dim, dtype=torch.bool)
            host[self.positive_dims] = True
            m = self._masks[like.device] = host.to(like.device)
        return m

    def to_state(self, z: Tensor) -> Tensor:
        if not self.positive_dims:
            return z
        return torch.where(self._mask(z), F.softplus(z), z)

    def to_latent(self, x: Tensor) -> Tensor:
        """Inverse softplus, x + log(1 - exp(-x)), with x clamped at 1e-6."""
        if not self.positive_dims:
            return x
        xp = x.clamp(min=1e-6)
        return torch.where(self._mask(x), xp + torch.log(-torch.expm1(-xp)), x)

    def log_jacobian(self, z: Tensor) -> Tensor:
        """log |dx/dz| summed over the state dimension = sum of logsigmoid(z) on positive dims."""
        if not self.positive_dims:
            return torch.zeros(z.shape[:-1], device=z.device, dtype=z.dtype)
        return F.logsigmoid(z[..., self.positive_dims]).sum(dim=-1)
