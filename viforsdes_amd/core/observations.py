"""Observations and the Gaussian observation model (reference: core/observations.py:12-74)."""
from __future__ import annotations

import math
import weakref
from typing import Optional, Protocol, runtime_checkable

import torch
from pydantic import BaseModel, ConfigDict, model_validator
from torch import Tensor
from typing_extensions import Self


class Observations(BaseModel):
    """``times[T_obs]`` (sorted) and ``values[T_obs, obs_dim]``."""

    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)
    times: Tensor
    values: Tensor

    @model_validator(mode="after")
    def _check(self) -> Self:
        t, v = self.times, self.values
        if t.ndim != 1:
            raise ValueError("times must be 1D tensor")
        if v.ndim != 2:
            raise ValueError("values must be 2D tensor [T_obs, obs_dim]")
        if t.shape[0] != v.shape[0]:
            raise ValueError(f"times and values must have same first dimension: got {t.shape[0]} vs {v.shape[0]}")
        if t.numel() > 1 and bool((t[1:] < t[:-1]).any()):
            raise ValueError("times must be sorted in non-decreasing order")
        return self

    def to(self, device: torch.device | str) -> "Observations":
        return Observations(times=self.times.to(device), values=self.values.to(device))


_GRID_INDEX_CACHE: dict = {}


def grid_index(times: Tensor, time_step: float, max_index: int) -> Tensor:
    """``round(times / time_step).long().clamp(max=max_index)``: the grid rows of the observation times (reference:
    evidence_lower_bound.py:45, encoder.py:74).  The observation times do not change during a run, so the index tensor is built
    once per (tensor, version, step, bound) instead of four small kernels at each of its two uses in every step."""
    key = (id(times), times._version, float(time_step), int(max_index))
    hit = _GRID_INDEX_CACHE.get(key)
    if hit is not None and hit[0]() is times:     # the id of a dead tensor can be reused: the weak reference pins the identity
        return hit[1]
    idx = torch.round(times / time_step).long().clamp(max=max_index)
    if times.is_cuda and torch.cuda.is_current_stream_capturing():
        return idx   # memory of a capturing graph's private pool: valid for that graph only
    if len(_GRID_INDEX_CACHE) >= 16:
        _GRID_INDEX_CACHE.clear()
    _GRID_INDEX_CACHE[key] = (weakref.ref(times), idx)
    return idx


@runtime_checkable
class ObservationLikelihood(Protocol):
    def log_prob(self, observations: Tensor, state: Tensor) -> Tensor: ...


class GaussianObservationLikelihood(BaseModel):
    """y ~ N(H x, variance * I); ``obs_matrix`` H is optional (identity when absent)."""

    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)
    variance: float
    obs_matrix: Optional[Tensor] = None

    @model_validator(mode="after")
    def _check(self) -> Self:
        if self.variance <= 0:
            raise ValueError("variance must be positive")
        return self

    def predict(self, state: Tensor) -> Tensor:
        H = self.obs_matrix
        if H is None:
            return state
        if H.ndim != 2:
            raise ValueError("obs_matrix must be 2D [obs_dim, state_dim]")
        if H.shape[1] != state.shape[-1]:
            raise ValueError("obs_matrix second dim must match state")
        return state @ H.to(state).T

    def log_prob(self, observations: Tensor, state: Tensor) -> Tensor:
        pred = self.predict(state)
        if self.obs_matrix is not None and self.obs_matrix.shape[0] != observations.shape[-1]:
            raise ValueError("obs_matrix first dim must match observations")
        if observations.shape != pred.shape:
            raise ValueError(f"observation shape {observations.shape} does not match predicted shape {pred.shape}")
        resid = observations - pred
        per_dim = -0.5 * resid * resid / self.variance - 0.5 * math.log(2.0 * math.pi * self.variance)
        return per_dim.sum(dim=-1)
