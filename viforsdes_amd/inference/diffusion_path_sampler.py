"""Encoder -> noise -> fused head (reference: inference/diffusion_path_sampler.py:35-69)."""
from __future__ import annotations

import os
from typing import Optional, Protocol

import torch
from torch import Tensor, nn

from ..core.observations import Observations
from .state_space import StateSpace
from .types import DiffusionPathSample


class EncoderProtocol(Protocol):
    def __call__(self, obs_values: Tensor, obs_times: Tensor, sde_parameters: Tensor, time_horizon: float,
                 time_step: float) -> Tensor: ...


class HeadProtocol(Protocol):
    def sample_diffusion_paths(self, x0: Tensor, context: Tensor, sde_parameters: Tensor, standard_noise: Tensor,
                               time_step: float) -> tuple[Tensor, Tensor, Tensor]: ...


def _latent_start(state_space: StateSpace, x0: Tensor) -> Tensor:
    """``state_space.to_latent(x0)`` (reference line 61), remembered while ``x0`` stays the same tensor: the start state is a
    constant of the problem (the trainer's ``x0_buffer``, the posterior's first observation), its inverse softplus seven small
    launches per call.  Recomputed when the memory, its version counter (an in-place edit) or its layout changes, when it
    carries a gradient, and inside a stream capture (a graph's private memory must not outlive it)."""
    if x0.requires_grad or (x0.is_cuda and torch.cuda.is_current_stream_capturing()):
        return state_space.to_latent(x0)
    key = (x0.data_ptr(), x0._version, tuple(x0.shape), tuple(x0.stride()), x0.dtype, x0.device)
    cache = state_space.__dict__.setdefault("_latent_start_cache", {})
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= 4:
            cache.clear()
        with torch.no_grad():
            # the entry keeps x0 (hence its storage) alive: the address in the key cannot be handed to another tensor meanwhile
            hit = cache[key] = (x0, state_space.to_latent(x0))
    return hit[1]


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
    z0 = _latent_start(state_space, x0)
    if getattr(head, "accepts_full_context", False):  # our head: reads the first T steps in place, gradient written in place
        paths, means, chol = head.sample_diffusion_paths(z0, context, sde_parameters, noise, time_step,
                                                         context_has_extra_step=True)
    else:
        paths, means, chol = head.sample_diffusion_paths(z0, context[:, :-1], sde_parameters, noise, time_step)
    return DiffusionPathSample(z=paths, transition_means=means, transition_cholesky=chol, state_space=state_space)


class CapturedPathSampler:
    """The no-grad sampling call (theta draw -> encoder -> noise -> fused head -> state transform; reference:
    posterior/variational_posterior.py:93-107 around diffusion_path_sampler.py:35-69) for a FIXED number of paths as one
    replayable HIP graph.

    A sampling call is ~250 launches; at the Ornstein-Uhlenbeck size (12.9 k encoder tokens) their host cost, not the GPU, sets
    the rate (2.5 ms eager for 0.8 ms of kernels).  ``sampler()`` replays the captured call -- fresh theta and noise draws every
    time (graph-safe Philox offsets) -- and returns ``(theta [n, P], x [n, T + 1, S], DiffusionPathSample)`` in STATIC buffers
    that the next call overwrites (clone what has to survive).  The graph reads the model's parameters in place: weights may
    change between calls (an EMA swap, an optimizer step); the bf16 operand packs of the encoder are brought up to date before
    every replay, outside the graph.  ``autocast_dtype``: run the encoder under autocast (the training step's precision) instead
    of the parameters' own.  Raises when capture is not possible (CPU, or a launch the capture rejects): callers keep the eager
    ``sample_diffusion_paths`` for that case."""

    def __init__(self, model: nn.Module, observations: Observations, time_horizon: float, time_step: float,
                 state_space: StateSpace, n: int, autocast_dtype: Optional[torch.dtype] = None, warmup: int = 2) -> None:
        from ..primitives.fused import PackedWeight
        dev = observations.values.device
        if dev.type != "cuda":
            raise RuntimeError("CapturedPathSampler needs the GPU path (HIP graph capture)")
        self._packs, self._ids = PackedWeight, {id(p) for p in model.parameters()}
        x0 = observations.values[0].unsqueeze(0).expand(n, -1)

        def draw():
            theta = model.sde_parameter_posterior.rsample(n)
            with torch.autocast(device_type="cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
                s = sample_diffusion_paths(model.encoder, model.head, observations, theta, x0, time_horizon, time_step,
                                           state_space)
            return theta, s.x, s

        rng = torch.cuda.get_rng_state(dev)   # warm-up draws must not show: the first replay continues the caller's stream
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):   # lazily built caches, allocator pools, operand packs
                    draw()
            torch.cuda.current_stream(dev).wait_stream(side)
            self._packs.refresh_all(params=self._ids)
            self.graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(self.graph):
                    self._static = draw()
            except Exception:
                torch.cuda.synchronize(dev)
                raise
            finally:
                torch.cuda.set_rng_state(rng, dev)
        self.n = n

    def __call__(self) -> tuple[Tensor, Tensor, DiffusionPathSample]:
        # nothing unless a parameter changed since the last call; then ONE kernel over all packs of this model (an EMA swap makes
        # every pack stale: the per-pack copies of the unforced refresh would be ~50 launches)
        if any(pk.stale() for pk in self._packs._live if any(id(q) in self._ids for q in pk.params)):
            self._packs.refresh_all(force=True, params=self._ids)
        # tile images derived from the packs: rebuilt here when dirty (a forced refresh, a replayed training step) -- the graph
        # reads them in place
        self._packs.refresh_dirty_derived(self._ids)
        self.graph.replay()
        return self._static

    @staticmethod
    def kernel_choice_outdated() -> bool:
        """The captured call keeps the GRU kernels it was captured with.  Once a head weight has left the f16 range of the multi-path
        MFMA kernels (sticky flag, ``_hip.head_mfma_range_exceeded``) a sampler captured before that must be re-captured: eager
        launches already take the fp32 kernels."""
        from .. import _hip
        return _hip.head_mfma_range_exceeded()


# VSDE_SAMPLE_GRAPH=0: VariationalPosterior.sample never replays a captured call (A/B runs, debugging)
SAMPLE_GRAPH = os.environ.get("VSDE_SAMPLE_GRAPH", "1") != "0"
