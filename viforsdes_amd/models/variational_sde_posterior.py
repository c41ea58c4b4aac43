"""Container whose ``state_dict()`` is the checkpoint layout
(reference: models/variational_sde_posterior.py:11-36): ``encoder.*``, ``head.*``,
``sde_parameter_posterior.*``."""
from __future__ import annotations

from typing import Optional

from torch import Tensor, nn

from ..config import EncoderConfig, HeadConfig
from .encoder import ObservationContextEncoder
from .head import DiffusionTransitionHead
from .sde_parameter_posterior import SDEParameterPosterior

_COMPILED_PREFIX = "encoder.sit._orig_mod."


class VariationalSDEPosterior(nn.Module):
    def __init__(self, observation_dim: int, state_dim: int, sde_param_dim: int, encoder_config: EncoderConfig,
                 head_config: HeadConfig, sde_param_positive_dims: list[int],
                 sde_param_init_mean: Optional[Tensor] = None) -> None:
        super().__init__()
        self.encoder = ObservationContextEncoder(observation_dim, sde_param_dim, encoder_config)
        self.head = DiffusionTransitionHead(state_dim, encoder_config.hidden_dim, sde_param_dim, head_config)
        self.sde_parameter_posterior = SDEParameterPosterior(sde_param_dim, sde_param_positive_dims,
                                                             init_mean=sde_param_init_mean)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Also accepts checkpoints written by the reference with ``torch.compile`` enabled, whose
        trunk keys are spelled ``encoder.sit._orig_mod.*``."""
        if any(k.startswith(_COMPILED_PREFIX) for k in state_dict):
            state_dict = {k.replace(_COMPILED_PREFIX, "encoder.sit.", 1): v for k, v in state_dict.items()}
        return super().load_state_dict(state_dict, strict=strict, assign=assign)
