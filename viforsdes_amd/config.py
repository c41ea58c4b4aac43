"""Configuration objects (reference: config.py:12-115).  Frozen pydantic models with the same
field names, defaults and validation behaviour; ``from_yaml`` loads a flat mapping."""
from __future__ import annotations

from enum import Enum
from pathlib import Path
from typing import Any

import torch
import yaml
from pydantic import BaseModel, ConfigDict, field_validator, model_validator
from typing_extensions import Self


def _positive(name: str, value: Any) -> Any:
    if value <= 0:
        raise ValueError(f"{name} must be positive, got {value}")
    return value


class YamlConfig(BaseModel):
    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)

    @classmethod
    def from_yaml(cls, path: str | Path) -> Self:
        text = Path(path).read_text()
        return cls(**(yaml.safe_load(text) or {}))


class AmpDtype(Enum):
    FLOAT16 = torch.float16
    BFLOAT16 = torch.bfloat16


class TrainingConfig(YamlConfig):
    time_step: float = 0.1
    batch_size: int = 50
    n_iterations: int = 25000
    learning_rate: float = 1e-4
    sde_param_lr: float = 1e-3
    grad_clip_norm: float = 1.0
    amp_dtype: AmpDtype = AmpDtype.BFLOAT16

    @model_validator(mode="after")
    def _check(self) -> Self:
        for name in ("time_step", "batch_size", "n_iterations", "learning_rate", "sde_param_lr", "grad_clip_norm"):
            _positive(name, getattr(self, name))
        return self


class EncoderConfig(YamlConfig):
    hidden_dim: int = 128
    cond_dim: int = 128
    num_heads: int = 4
    depth: int = 4
    mlp_ratio: float = 8 / 3

    @model_validator(mode="after")
    def _check(self) -> Self:
        for name in ("hidden_dim", "cond_dim", "num_heads", "depth", "mlp_ratio"):
            _positive(name, getattr(self, name))
        if self.hidden_dim % self.num_heads != 0:
            raise ValueError("hidden_dim must be divisible by num_heads")
        return self


class HeadConfig(YamlConfig):
    hidden_dim: int = 64
    num_layers: int = 2

    @field_validator("hidden_dim", "num_layers")
    @classmethod
    def _check(cls, v: int) -> int:
        return _positive("value", v)


class PretrainConfig(YamlConfig):
    n_iterations: int = 1000
    batch_size: int = 4096
    learning_rate: float = 0.02
    init_scale: float = 2.0

    @model_validator(mode="after")
    def _check(self) -> Self:
        for name in ("n_iterations", "batch_size", "learning_rate", "init_scale"):
            _positive(name, getattr(self, name))
        return self
