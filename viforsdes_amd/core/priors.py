"""iid Normal / LogNormal prior over the SDE parameters (reference: core/priors.py:19-60)."""
from __future__ import annotations

import math
from enum import Enum, auto

import torch
from pydantic import BaseModel, ConfigDict, model_validator
from torch import Tensor
from typing_extensions import Self

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class PriorType(Enum):
    NORMAL = auto()
    LOG_NORMAL = auto()


class Prior(BaseModel):
    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)
    type: PriorType
    mean: float
    std: float
    dim: int

    @model_validator(mode="after")
    def _check(self) -> Self:
        if self.dim <= 0:
            raise ValueError("dim must be positive")
        if self.std <= 0:
            raise ValueError("std must be positive")
        return self

    def sample(self, n: int) -> Tensor:
        draw = self.mean + self.std * torch.randn(n, self.dim)
        return draw.exp() if self.type == PriorType.LOG_NORMAL else draw

    def log_prob(self, sde_parameters: Tensor) -> Tensor:
        """Sum over the parameter dimension of the per-coordinate log density -> ``[B]``.

        Closed form instead of rebuilding ``torch.distributions`` objects on every call
        (the reference does, core/priors.py:46-60; same value)."""
        if sde_parameters.shape[-1] != self.dim:
            raise ValueError(f"expected last dim {self.dim}, got {sde_parameters.shape[-1]}")
        if self.type == PriorType.LOG_NORMAL:
            logx = sde_parameters.log()
            zed = (logx - self.mean) / self.std
            lp = -0.5 * zed * zed - math.log(self.std) - _HALF_LOG_2PI - logx
        else:
            zed = (sde_parameters - self.mean) / self.std
            lp = -0.5 * zed * zed - math.log(self.std) - _HALF_LOG_2PI
        return lp.sum(dim=-1)
