"""viforsdes_amd -- MI355X-native variational inference for SDEs.

Same user-facing surface as Tom-Ryder/VIforSDEs (``infer``, ``InferenceConfig``, ``SDE``,
``Observations``, ``Prior``, ``GaussianObservationLikelihood``, the configs and the
``VariationalSDEPosterior`` checkpoint layout); the hot path (fused GRU path sampler forward /
backward and the ELBO accumulation) runs as hand-written HIP kernels for gfx950 behind the C ABI
declared in ``include/vsde_hip.h``."""
import os as _os

# dmabuf IPC for multi-process GPU work (RCCL): read once when HSA initialises, so it is defaulted here, at import time
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .config import AmpDtype, EncoderConfig, HeadConfig, PretrainConfig, TrainingConfig, YamlConfig
from .core.observations import GaussianObservationLikelihood, ObservationLikelihood, Observations
from .core.priors import Prior, PriorType
from .core.sde import SDE, FunctionalSDE, make_sde
from .infer import InferenceConfig, infer
from .posterior.variational_posterior import VariationalPosterior

__all__ = ["AmpDtype", "EncoderConfig", "HeadConfig", "PretrainConfig", "TrainingConfig", "YamlConfig",
           "GaussianObservationLikelihood", "ObservationLikelihood", "Observations", "Prior", "PriorType", "SDE",
           "FunctionalSDE", "make_sde", "InferenceConfig", "infer", "VariationalPosterior"]
__version__ = "0.1.0"
