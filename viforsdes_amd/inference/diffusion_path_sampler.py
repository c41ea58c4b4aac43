"""Encoder -> noise -> fused head (reference: inference/diffusion_path_sampler.py:35-69)."""
from __future__ import annotations

from typing import Optional, Protocol

import torch
from torch import Tensor

from ..core.observations import Observations
from .state_space import StateSpace
from .types import DiffusionPathSample


class EncoderProtocol(Protocol):
    def __call__(self, obs_values: Tensor, obs_times: Tensor, sde_parameters: Tensor, time_horizon: float,
                 time_step: float) -> Tensor: ...


class HeadProtocol(Protocol):
    def sample_diffusion_paths(self, x0: Tensor, context: Tensor, sde_parameters: Tensor, standard_noise: Tensor,
                               time_step: float) -> tuple[Tensor, Tensor, Tensor]: ...


def sample_diffusion_paths(encoder: EncoderProtocol, head: HeadProtocol, observations: Observations,
                           sde_parameters: Tensor, x0: Tensor, time_horizon: float, time_step: float,
                           state_space: StateSpace, noise: Optional[Tensor] = None) -> DiffusionPathSample:
    """Draw one latent path per row of ``sde_parameters``.

    ``noise`` (``[B, T, S]`` standard normal) may be injected for reproducibility; by default it is
    drawn here, after the caller's theta draw -- the same RNG order as the reference (line 57)."""
    B, S = x0.shape
    context = encoder(observations.values, observations.times, sde_parameters, time_horizon, time_step)
    n_steps = context.shape[1] - 1
    if noise is None:
        noise = torch.randn(B, n_steps, S, device=x0.device, dtype=x0.dtype)
    elif tuple(noise.shape) != (B, n_steps, S):
        raise ValueError(f"noise must have shape {(B, n_steps, S)}, got {tuple(noise.shape)}")
    z0 = state_space.to_latent(x0)
    if getattr(head, "accepts_full_context", False):  # our head: reads the first T steps in place, gradient written in place
        paths, means, chol = head.sample_diffusion_paths(z0, context, sde_parameters, noise, time_step,
                                                         context_has_extra_step=True)
    else:
        paths, means, chol = head.sample_diffusion_paths(z0, context[:, :-1], sde_parameters, noise, time_step)
    return DiffusionPathSample(z=paths, transition_means=means, transition_cholesky=chol, state_space=state_space)
