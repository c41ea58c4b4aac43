"""Result object of ``infer`` (reference: posterior/variational_posterior.py:23-192): sampling with
the EMA weights, summaries, diagnostics and the on-disk checkpoint

    {model_state, ema_state, time_horizon, time_step, state_positive_dims, evidence_lower_bound_history}

which is byte-compatible with the reference's ``torch.save`` layout (same keys, same tensors)."""
from __future__ import annotations

import os

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import torch
from pydantic import BaseModel, ConfigDict
from torch import Tensor

from ..core.observations import Observations
from ..core.priors import Prior
from ..inference import diffusion_path_sampler as _sampler
from ..inference.diffusion_path_sampler import CapturedPathSampler, sample_diffusion_paths
from ..inference.exponential_moving_average import ExponentialMovingAverage
from ..inference.state_space import StateSpace
from ..models.variational_sde_posterior import VariationalSDEPosterior

QUANTILE_LEVELS = (0.05, 0.25, 0.5, 0.75, 0.95)


@dataclass(frozen=True)
class VariationalPosteriorSamples:
    sde_parameters: Tensor
    diffusion_paths: Tensor


@dataclass(frozen=True)
class Quantiles:
    q05: Tensor
    q25: Tensor
    q50: Tensor
    q75: Tensor
    q95: Tensor


@dataclass
class VariationalPosteriorSummary:
    sde_parameter_mean: Tensor
    sde_parameter_std: Tensor
    sde_parameter_quantiles: Quantiles
    diffusion_path_mean: Tensor
    diffusion_path_std: Tensor


@dataclass
class InferenceDiagnostics:
    evidence_lower_bound_history: list[float]
    final_evidence_lower_bound: float
    n_iterations: int


class VariationalPosteriorCheckpoint(BaseModel):
    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)
    model_state: dict[str, Tensor]
    ema_state: dict[str, Tensor]
    time_horizon: float
    time_step: float
    state_positive_dims: list[int]
    evidence_lower_bound_history: list[float]


class VariationalPosterior:
    def __init__(self, model: VariationalSDEPosterior, exponential_moving_average: ExponentialMovingAverage,
                 prior: Prior, observations: Observations, time_horizon: float, time_step: float,
                 state_space: StateSpace, evidence_lower_bound_history: list[float], device: torch.device) -> None:
        self.model = model.to(device)
        self.exponential_moving_average = exponential_moving_average
        self.prior, self.observations = prior, observations.to(device)
        self.time_horizon, self.time_step, self.state_space = time_horizon, time_step, state_space
        self.evidence_lower_bound_history = evidence_lower_bound_history
        self.device = device
        self._calls: dict[tuple, int] = {}                                  # sample() calls seen per (n, autocast dtype)
        self._captured: dict[tuple, Optional[CapturedPathSampler]] = {}     # -> replayable call (None: capture failed, stay eager)
        self._capture_failure_logged = False
        # sample(n) is replayed from a HIP graph from its second call on, for n up to this many samples (a graph pins the call's
        # peak memory: ~0.5 GB at 512 Lotka-Volterra paths); 0 = always eager.  release_graphs() frees the pools.
        self.graph_max_samples = int(os.environ.get("VSDE_SAMPLE_GRAPH_MAX", "1024"))

    @torch.no_grad()
    def sample(self, n: int, mixed_precision: bool = False) -> VariationalPosteriorSamples:
        """n joint draws (theta, path) from the variational posterior using the EMA weights.

        On the GPU a repeated ``sample(n)`` with the same ``n`` (a serving loop) replays the call as ONE HIP graph from its
        second occurrence on (``CapturedPathSampler``: same kernels, same RNG stream order -- theta draw, then the path noise --
        fresh draws per call; ``VSDE_SAMPLE_GRAPH=0`` keeps every call eager).  The first call of a size runs eagerly.
        ``mixed_precision`` (not in the reference, whose ``sample`` always runs in the parameters' precision): run the encoder
        under bf16 autocast, i.e. in the precision a ``mixed_precision`` training run optimised it in -- the fused encoder
        kernels instead of fp32 library GEMMs (~5 x faster at the Lotka-Volterra size); the head stays fp32."""
        self.model.eval()
        amp = torch.bfloat16 if (mixed_precision and self.device.type == "cuda") else None
        with self.exponential_moving_average.apply():
            replay = self._replayable(n, amp)
            if replay is not None:
                theta, x, _ = replay()
                return VariationalPosteriorSamples(sde_parameters=theta.clone(), diffusion_paths=x.clone())
            theta = self.model.sde_parameter_posterior.rsample(n)
            x0 = self.observations.values[0].unsqueeze(0).expand(n, -1)
            with torch.autocast(device_type=self.device.type, dtype=amp, enabled=amp is not None):
                drawn = sample_diffusion_paths(self.model.encoder, self.model.head, self.observations, theta, x0,
                                               self.time_horizon, self.time_step, self.state_space)
            return VariationalPosteriorSamples(sde_parameters=theta, diffusion_paths=drawn.x)

    def _replayable(self, n: int, amp: Optional[torch.dtype] = None) -> Optional[CapturedPathSampler]:
        if self.device.type != "cuda" or not _sampler.SAMPLE_GRAPH:
            return None
        if self._captured and not getattr(self, "_range_flag_seen", False) and CapturedPathSampler.kernel_choice_outdated():
            self._range_flag_seen = True   # captured with the MFMA GRU kernels, which a weight has outgrown: capture again (fp32 kernels)
            self._captured.clear()
        key = (n, amp)
        self._calls[key] = self._calls.get(key, 0) + 1
        if self._calls[key] < 2:
            return None
        if n > self.graph_max_samples:     # a captured call pins the peak memory of one sampling call at this size for good
            return None
        if key not in self._captured:
            if len(self._captured) >= 2:   # each graph keeps a private memory pool: two sizes at most (oldest goes first)
                self._captured.pop(next(iter(self._captured)))
            try:
                self._captured[key] = CapturedPathSampler(self.model, self.observations, self.time_horizon, self.time_step,
                                                          self.state_space, n, autocast_dtype=amp, warmup=1)
            except Exception as err:   # capture is an optimisation, never a requirement -- but a persistent fallback must be visible
                self._captured[key] = None
                if not self._capture_failure_logged:
                    self._capture_failure_logged = True
                    import logging
                    logging.getLogger("viforsdes_amd").warning(
                        "VariationalPosterior.sample(%d): HIP graph capture failed (%s: %s); sampling eagerly", n, type(err).__name__, err)
        return self._captured[key]

    def release_graphs(self) -> None:
        """Drop the captured sampling calls and their private memory pools (they are re-captured on demand)."""
        self._captured.clear()
        self._calls.clear()
        if self.device.type == "cuda":
            torch.cuda.empty_cache()

    def summary(self, n_samples: int = 1000, mixed_precision: bool = False) -> VariationalPosteriorSummary:
        s = self.sample(n_samples, mixed_precision)
        levels = torch.tensor(QUANTILE_LEVELS, device=self.device, dtype=s.sde_parameters.dtype)
        q = torch.quantile(s.sde_parameters, levels, dim=0)
        return VariationalPosteriorSummary(
            sde_parameter_mean=s.sde_parameters.mean(dim=0), sde_parameter_std=s.sde_parameters.std(dim=0),
            sde_parameter_quantiles=Quantiles(*q.unbind(0)), diffusion_path_mean=s.diffusion_paths.mean(dim=0),
            diffusion_path_std=s.diffusion_paths.std(dim=0))

    def diagnostics(self) -> InferenceDiagnostics:
        hist = self.evidence_lower_bound_history
        return InferenceDiagnostics(evidence_lower_bound_history=hist,
                                    final_evidence_lower_bound=hist[-1] if hist else float("nan"),
                                    n_iterations=len(hist))

    def plot(self, n_trajectories: int = 50, show: bool = True):
        from ..visualization import plot_posterior
        return plot_posterior(self.sample(n_trajectories), self.observations, self.time_horizon, show)

    def save(self, path: str | Path) -> None:
        torch.save({"model_state": self.model.state_dict(),
                    "ema_state": self.exponential_moving_average.state_dict(),
                    "time_horizon": self.time_horizon, "time_step": self.time_step,
                    "state_positive_dims": self.state_space.positive_dims,
                    "evidence_lower_bound_history": self.evidence_lower_bound_history}, Path(path))

    @classmethod
    def load(cls, path: str | Path, model: VariationalSDEPosterior, prior: Prior, observations: Observations,
             device: torch.device) -> "VariationalPosterior":
        ckpt = VariationalPosteriorCheckpoint.model_validate(torch.load(Path(path), map_location=device, weights_only=True))
        model.load_state_dict(ckpt.model_state)
        ema = ExponentialMovingAverage(model)
        ema.load_state_dict(ckpt.ema_state)
        space = StateSpace(observations.values.shape[-1], ckpt.state_positive_dims)
        return cls(model=model, exponential_moving_average=ema, prior=prior, observations=observations,
                   time_horizon=ckpt.time_horizon, time_step=ckpt.time_step, state_space=space,
                   evidence_lower_bound_history=ckpt.evidence_lower_bound_history, device=device)
