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


def builtin_sde_kind(sde: object) -> str | None:
    """The HIP library's name for ``sde`` when -- and only when -- its drift and diffusion are exactly the closed forms the
    library implements (csrc/vsde_sde.hip), else None (the Python callables are then used).

    ``builtin_kind`` is a class attribute of the example SDEs and is therefore inherited: a subclass that overrides
    ``drift`` or ``diffusion`` (or an instance that shadows them) must NOT be routed to the built-in kernels, which would
    silently ignore the override.  The dispatch holds only if both callables still resolve to the functions of the class
    that declared ``builtin_kind``, and the dimensions are inside what the kernels accept (linear-diagonal: state_dim <= 32)."""
    cls = type(sde)
    owner = next((c for c in cls.__mro__ if "builtin_kind" in vars(c)), None)
    if owner is None or "builtin_kind" in getattr(sde, "__dict__", {}):
        return None
    kind = vars(owner)["builtin_kind"]
    for name in ("drift", "diffusion"):
        if name in getattr(sde, "__dict__", {}) or getattr(cls, name, None) is not vars(owner).get(name):
            return None
    if kind == "linear_diagonal" and not (1 <= int(getattr(sde, "state_dim", 0)) <= 32):
        return None
    return kind
