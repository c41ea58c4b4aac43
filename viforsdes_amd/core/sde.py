"""The user-facing SDE definition (reference: core/sde.py:8-48).

``drift(x[N,S], theta[N,P]) -> [N,S]``; ``diffusion(x, theta) -> [N,S,S]`` must be a
lower-triangular factor with positive diagonal (it is used as a Cholesky factor of the
transition covariance, reference evidence_lower_bound.py:43,81)."""
from __future__ import annotations

from typing import Callable, Protocol, runtime_checkable

from torch import Tensor

TensorFn = Callable[[Tensor, Tensor], Tensor]


@runtime_checkable
class SDE(Protocol):
    state_dim: int
    sde_param_dim: int

    def drift(self, x: Tensor, sde_parameters: Tensor) -> Tensor: ...

    def diffusion(self, x: Tensor, sde_parameters: Tensor) -> Tensor: ...


class FunctionalSDE:
    """Adapter turning two callables into an :class:`SDE`."""

    def __init__(self, drift_fn: TensorFn, diffusion_fn: TensorFn, state_dim: int, sde_param_dim: int) -> None:
        self.state_dim = state_dim
        self.sde_param_dim = sde_param_dim
        self._f, self._g = drift_fn, diffusion_fn

    def drift(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return self._f(x, sde_parameters)

    def diffusion(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return self._g(x, sde_parameters)


def make_sde(drift: TensorFn, diffusion: TensorFn, state_dim: int, sde_param_dim: int) -> SDE:
    return FunctionalSDE(drift, diffusion, state_dim, sde_param_dim)
