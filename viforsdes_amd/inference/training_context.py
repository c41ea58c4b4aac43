"""Everything one training run owns (reference: inference/training_context.py:23-135): model,
AdamW with two parameter groups, GradScaler, EMA, device copies of the observations, the x0
buffer, and -- when launched under torchrun -- the process group and the flat gradient exchange."""
from __future__ import annotations

from dataclasses import dataclass
from typing import TYPE_CHECKING, Optional

import torch
import torch.distributed as dist
from torch.amp import GradScaler

from ..config import EncoderConfig, HeadConfig, TrainingConfig
from ..core.observations import Observations
from ..models.variational_sde_posterior import VariationalSDEPosterior
from .data_parallel import FlatGradientAllReduce, broadcast_module_state, env_rank_info, init_process_group_if_needed
from .exponential_moving_average import ExponentialMovingAverage

if TYPE_CHECKING:
    from ..accelerate import Accelerator

DEFAULT_DISTRIBUTED_SEED = 1234


@dataclass
class TrainingContext:
    model: VariationalSDEPosterior
    optimizer: torch.optim.AdamW
    scaler: GradScaler
    ema: ExponentialMovingAverage
    observations: Observations
    x0_buffer: torch.Tensor
    device: torch.device
    is_distributed: bool
    is_main: bool
    local_rank: int
    rank: int
    world_size: int
    grad_sync: FlatGradientAllReduce

    def trainable_model(self) -> VariationalSDEPosterior:
        return self.model

    def unwrap_model(self) -> VariationalSDEPosterior:
        return self.model

    @classmethod
    def create(cls, observations: Observations, state_dim: int, sde_param_dim: int, config: TrainingConfig,
               encoder_config: EncoderConfig, head_config: HeadConfig, sde_param_positive_dims: list[int],
               device: torch.device | str, mixed_precision: bool, accelerator: "Optional[Accelerator]",
               sde_param_init_mean: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> "TrainingContext":
        rank, local_rank, world = env_rank_info()
        dev = torch.device(device) if isinstance(device, str) else device
        distributed = world > 1
        if distributed:
            if dev.type == "cuda":
                dev = torch.device(f"cuda:{local_rank}")
                torch.cuda.set_device(dev)
            init_process_group_if_needed(dev.type)
        if dev.type == "cuda":
            from ..accelerate import enable_tuned_gemms
            enable_tuned_gemms()
        if seed is None and distributed:
            seed = DEFAULT_DISTRIBUTED_SEED
        if seed is not None:
            torch.manual_seed(seed)  # identical initialisation on every rank ...

        model = VariationalSDEPosterior(
            observation_dim=observations.values.shape[-1], state_dim=state_dim, sde_param_dim=sde_param_dim,
            encoder_config=encoder_config, head_config=head_config, sde_param_positive_dims=sde_param_positive_dims,
            sde_param_init_mean=sde_param_init_mean).to(dev)
        if accelerator is not None:
            model.encoder.sit = accelerator.optimize(model.encoder.sit)
        if distributed:
            broadcast_module_state(model, src=0)
        if seed is not None:
            torch.manual_seed(seed + rank)  # ... independent Monte-Carlo draws per rank
        ema = ExponentialMovingAverage(model)

        theta_params = list(model.sde_parameter_posterior.parameters())
        theta_ids = {id(p) for p in theta_params}
        net_params = [p for p in model.parameters() if id(p) not in theta_ids]
        # Same AdamW hyper-parameters as the reference (training_context.py:97-102).  On the GPU the fused
        # implementation is used: GradScaler hands it `found_inf` on the device, so the optimizer step needs
        # no host synchronisation (the unfused path does a .item() per step, which drains the launch queue);
        # capturable=True keeps the step counters on the device so the whole step can be captured in a HIP graph.
        optimizer = torch.optim.AdamW([{"params": net_params, "lr": config.learning_rate},
                                       {"params": theta_params, "lr": config.sde_param_lr}],
                                      fused=dev.type == "cuda", capturable=dev.type == "cuda")
        scaler = GradScaler("cuda", enabled=bool(mixed_precision and dev.type == "cuda" and torch.cuda.is_available()))
        dev_obs = observations.to(dev)
        x0 = dev_obs.values[0].unsqueeze(0).expand(config.batch_size, -1).contiguous()
        return cls(model=model, optimizer=optimizer, scaler=scaler, ema=ema, observations=dev_obs, x0_buffer=x0,
                   device=dev, is_distributed=distributed, is_main=rank == 0, local_rank=local_rank, rank=rank,
                   world_size=world, grad_sync=FlatGradientAllReduce(model.parameters()))

    def cleanup(self) -> None:
        if self.is_distributed and dist.is_initialized():
            dist.destroy_process_group()
