"""Ornstein-Uhlenbeck and Lotka-Volterra model SDEs with the observation sets of the reference's
example scripts (examples/ornstein_uhlenbeck.py:18-51, examples/lotka_volterra.py:18-70).  Used by
bench.py, the smoke test and the example runners."""
from __future__ import annotations

import torch
from torch import Tensor

from ..core.observations import GaussianObservationLikelihood, Observations
from ..core.priors import Prior, PriorType


class OrnsteinUhlenbeck:
    """dx = kappa (mu - x) dt + sigma dW;  theta = (kappa, mu, sigma)."""

    state_dim = 1
    sde_param_dim = 3
    builtin_kind = "ornstein_uhlenbeck"  # drift / diffusion also exist inside the HIP simulator (csrc/vsde_sde.hip)

    def drift(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return sde_parameters[..., 0:1] * (sde_parameters[..., 1:2] - x)

    def diffusion(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return sde_parameters[..., 2:3].reshape(x.shape[0], 1, 1)


class LotkaVolterra:
    """Predator-prey diffusion approximation; theta = (theta1, theta2, theta3), state (u, v).

    Drift (t1 u - t2 u v, t2 u v - t3 v); diffusion = analytic 2x2 Cholesky factor of
    [[t1 u + t2 u v, -t2 u v], [-t2 u v, t3 v + t2 u v]] with 1e-6 floors."""

    state_dim = 2
    sde_param_dim = 3
    builtin_kind = "lotka_volterra"

    def drift(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        u, v = x.unbind(-1)
        t1, t2, t3 = sde_parameters.unbind(-1)
        uv = t2 * u * v
        return torch.stack([t1 * u - uv, uv - t3 * v], dim=-1)

    def diffusion(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        u, v = x.unbind(-1)
        t1, t2, t3 = sde_parameters.unbind(-1)
        uv = t2 * u * v
        l00 = torch.sqrt((t1 * u + uv).clamp(min=1e-6))
        l10 = -uv / l00.clamp(min=1e-6)
        l11 = torch.sqrt((t3 * v + uv - l10 * l10).clamp(min=1e-6))
        zero = torch.zeros_like(l00)
        return torch.stack([torch.stack([l00, zero], dim=-1), torch.stack([l10, l11], dim=-1)], dim=-2)


class LinearDiagonalSDE:
    """Synthetic stress workload (BASELINE config 5): dx = -a * x dt + diag(softplus(b)) dW,
    theta = (a[S], b[S])."""

    builtin_kind = "linear_diagonal"

    def __init__(self, state_dim: int = 8) -> None:
        self.state_dim = state_dim
        self.sde_param_dim = 2 * state_dim

    def drift(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return -sde_parameters[..., :self.state_dim] * x

    def diffusion(self, x: Tensor, sde_parameters: Tensor) -> Tensor:
        return torch.diag_embed(torch.nn.functional.softplus(sde_parameters[..., self.state_dim:]) + 1e-3)


def ou_problem():
    """(sde, observations, likelihood, prior, time_horizon, time_step, state_pos, theta_pos)."""
    obs = Observations(times=torch.tensor([0.0, 1.0, 2.0, 3.0, 4.0, 5.0]),
                       values=torch.tensor([[2.0], [1.5], [0.8], [1.2], [0.9], [1.1]]))
    return (OrnsteinUhlenbeck(), obs, GaussianObservationLikelihood(variance=0.1),
            Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3), 5.0, 0.05, [], [0, 2])


def lv_problem():
    obs = Observations(times=torch.tensor([0.0, 10.0, 20.0, 30.0, 40.0]),
                       values=torch.tensor([[71.0, 79.0], [47.61225908, 447.20971405], [80.53119269, 50.26254069],
                                            [23.10087379, 339.40432691], [158.05238324, 66.79611979]]))
    return (LotkaVolterra(), obs, GaussianObservationLikelihood(variance=1.0),
            Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3), 40.0, 0.1, [0, 1], [0, 1, 2])


def synthetic_problem(state_dim: int = 8):
    n_obs = 11
    obs = Observations(times=torch.linspace(0.0, 10.0, n_obs),
                       values=torch.sin(torch.arange(n_obs * state_dim, dtype=torch.float32)).reshape(n_obs, state_dim))
    sde = LinearDiagonalSDE(state_dim)
    return (sde, obs, GaussianObservationLikelihood(variance=0.25),
            Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=sde.sde_param_dim), 10.0, 0.01, [], [])
