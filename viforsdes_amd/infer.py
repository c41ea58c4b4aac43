"""``infer()``: the single public entry point (reference: infer.py:24-151)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import TYPE_CHECKING, Optional

import torch
from torch import Tensor

from .config import EncoderConfig, HeadConfig, PretrainConfig, TrainingConfig
from .core.observations import ObservationLikelihood, Observations
from .core.priors import Prior
from .core.sde import SDE
from .inference.state_space import StateSpace
from .inference.trainer import VariationalInferenceTrainer
from .posterior.variational_posterior import VariationalPosterior

if TYPE_CHECKING:
    from .accelerate import Accelerator
    from .console import Console


@dataclass(frozen=True)
class InferenceConfig:
    training: TrainingConfig = field(default_factory=TrainingConfig)
    encoder: EncoderConfig = field(default_factory=EncoderConfig)
    head: HeadConfig = field(default_factory=HeadConfig)
    state_positive_dims: list[int] = field(default_factory=list)
    sde_param_positive_dims: list[int] = field(default_factory=list)
    device: str | torch.device = "cuda"
    mixed_precision: bool = True
    param_names: Optional[list[str]] = None
    accelerator: "Optional[Accelerator]" = None
    sde_param_init_mean: Optional[Tensor] = None
    pretrain: bool | PretrainConfig = False
    console: "Optional[Console]" = None
    seed: Optional[int] = None  # additive: base RNG seed (rank r uses seed + r); None = unseeded single process


def validate_inference_inputs(observations: Observations, time_horizon: float, time_step: float, state_dim: int,
                              sde_param_dim: int, state_positive_dims: list[int], sde_param_positive_dims: list[int],
                              prior: Prior) -> None:
    """Grid/shape rules of the reference (infer.py:52-85); raises ``ValueError``."""
    if time_horizon <= 0:
        raise ValueError("time_horizon must be positive")
    if time_step <= 0:
        raise ValueError("time_step must be positive")
    times = observations.times
    if times.numel() == 0:
        raise ValueError("observations must be non-empty")
    ratio = time_horizon / time_step
    if not math.isclose(ratio, round(ratio), rel_tol=1e-6, abs_tol=1e-6):
        raise ValueError("time_horizon must be an integer multiple of time_step")
    tol = max(1e-6, 1e-4 * time_step)
    if abs(float(times[0])) > tol:
        raise ValueError("first observation time must be 0")
    if bool(((torch.round(times / time_step) * time_step - times).abs() > tol).any()):
        raise ValueError("observation times must align to time_step grid")
    if bool((times < 0).any()) or bool((times > time_horizon).any()):
        raise ValueError("observation times must be within [0, time_horizon]")
    for name, dims, bound in (("state_positive_dims", state_positive_dims, state_dim),
                              ("sde_param_positive_dims", sde_param_positive_dims, sde_param_dim)):
        if len(set(dims)) != len(dims):
            raise ValueError(f"{name} must be unique")
        if any(d < 0 or d >= bound for d in dims):
            raise ValueError(f"{name} must be within [0, {bound})")
    if prior.dim != sde_param_dim:
        raise ValueError("prior dim must match sde_param_dim")


def infer(sde: SDE, observations: Observations, observation_likelihood: ObservationLikelihood, prior: Prior,
          time_horizon: float, config: Optional[InferenceConfig] = None) -> VariationalPosterior:
    cfg = config or InferenceConfig()
    # Unlike the reference (infer.py:97) there is no silent CPU fallback: its own fallback cannot run
    # the fused head either (Triton has no CPU driver); here a missing GPU is an immediate error.
    device = cfg.device
    if str(device).startswith("cuda") and not torch.cuda.is_available():
        raise RuntimeError("infer(): device 'cuda' requested but no HIP device is available; the fused "
                           "head/ELBO kernels have no CPU fallback")
    state_pos, theta_pos = list(cfg.state_positive_dims), list(cfg.sde_param_positive_dims)
    validate_inference_inputs(observations, time_horizon, cfg.training.time_step, sde.state_dim, sde.sde_param_dim,
                              state_pos, theta_pos, prior)
    trainer = VariationalInferenceTrainer(
        sde=sde, observations=observations, observation_likelihood=observation_likelihood, prior=prior,
        time_horizon=time_horizon, config=cfg.training, encoder_config=cfg.encoder, head_config=cfg.head,
        state_positive_dims=state_pos, sde_param_positive_dims=theta_pos, device=device,
        mixed_precision=cfg.mixed_precision, console=cfg.console, param_names=cfg.param_names,
        accelerator=cfg.accelerator, sde_param_init_mean=cfg.sde_param_init_mean, seed=cfg.seed)
    if cfg.pretrain and cfg.sde_param_init_mean is None:
        pre_cfg = cfg.pretrain if isinstance(cfg.pretrain, PretrainConfig) else None
        best = trainer.pretrain_sde_parameters(pre_cfg)
        if trainer.ctx.is_distributed:
            # every rank pre-trains on its own draws (seed + rank); all of them must start the ELBO phase from ONE mean,
            # otherwise the replicas never agree again (gradient averaging does not remove a parameter offset)
            import torch.distributed as dist
            best = best.contiguous()
            dist.broadcast(best, src=0)
        with torch.no_grad():
            trainer.ctx.model.sde_parameter_posterior.mean.copy_(best)
            trainer.ctx.ema._init_shadow()
    try:
        state = trainer.train()
    finally:
        trainer.cleanup()
    return VariationalPosterior(
        model=state.model, exponential_moving_average=state.exponential_moving_average, prior=prior,
        observations=observations, time_horizon=time_horizon, time_step=cfg.training.time_step,
        state_space=StateSpace(sde.state_dim, state_pos),
        evidence_lower_bound_history=state.evidence_lower_bound_history, device=trainer.device)
