"""Mean-field variational posterior q(theta) (reference: models/sde_parameter_posterior.py:10-69):
Gaussian in the unconstrained space, exponentiated on the positive dimensions (=> LogNormal)."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class SDEParameterPosterior(nn.Module):
    positive_mask: Tensor

    def __init__(self, sde_param_dim: int, sde_param_positive_dims: list[int], init_mean: Optional[Tensor] = None,
                 init_std: float = 1.0) -> None:
        super().__init__()
        if sde_param_dim < 1:
            raise ValueError(f"sde_param_dim must be >= 1, got {sde_param_dim}")
        if init_std <= 0:
            raise ValueError(f"init_std must be positive, got {init_std}")
        if any(d < 0 or d >= sde_param_dim for d in sde_param_positive_dims):
            raise ValueError(f"sde_param_positive_dims must be in [0, {sde_param_dim})")
        self.sde_param_dim = sde_param_dim
        self._positive_cache: tuple[int, tuple[int, ...]] | None = None
        self.mean = nn.Parameter(torch.zeros(sde_param_dim) if init_mean is None else init_mean.clone())
        self.log_std = nn.Parameter(torch.full((sde_param_dim,), math.log(init_std)))
        mask = torch.zeros(sde_param_dim, dtype=torch.bool)
        mask[list(sde_param_positive_dims)] = True
        self.register_buffer("positive_mask", mask)

    @property
    def _positive_dims(self) -> tuple[int, ...]:
        """Host copy of ``positive_mask`` for the fused ELBO tail, re-derived whenever the buffer changes (``load_state_dict``
        or an in-place edit bump its version counter), so the kernel's mask is always the one ``rsample`` drew with."""
        mask = self.positive_mask
        key = (mask._version, id(mask))
        if self._positive_cache is None or self._positive_cache[0] != key:
            self._positive_cache = (key, tuple(int(i) for i in torch.nonzero(mask.detach().cpu()).flatten().tolist()))
        return self._positive_cache[1]

    def transform(self, unconstrained: Tensor) -> Tensor:
        return torch.where(self.positive_mask, unconstrained.exp(), unconstrained)

    def rsample(self, n: int, eps: Optional[Tensor] = None) -> Tensor:
        """Reparameterised draw ``[n, P]``; ``eps`` overrides the standard-normal noise (tests)."""
        if eps is None:
            eps = torch.randn(n, self.sde_param_dim, device=self.mean.device, dtype=self.mean.dtype)
        return self.transform(self.mean + self.log_std.exp() * eps)

    def log_prob(self, sde_parameters: Tensor) -> Tensor:
        pos = self.positive_mask
        safe = torch.where(pos, sde_parameters, torch.ones_like(sde_parameters))
        log_theta = safe.log()
        u = torch.where(pos, log_theta, sde_parameters)            # unconstrained coordinate
        zed = (u - self.mean) * torch.exp(-self.log_std)
        lp = -0.5 * zed * zed - self.log_std - _HALF_LOG_2PI - torch.where(pos, log_theta, torch.zeros_like(u))
        return lp.sum(dim=-1)

    @property
    def expected_value(self) -> Tensor:
        var = torch.exp(2.0 * self.log_std)
        return torch.where(self.positive_mask, torch.exp(self.mean + 0.5 * var), self.mean)
