"""Observation/context encoder (reference: models/encoder.py:16-99).

Builds one token per point of the (T+1)-point time grid -- a learned ``bridge_token`` everywhere,
``obs_proj(y_k)`` at the grid index of each observation, plus a sinusoidal time embedding -- and
runs the SiT trunk conditioned on ``MLP(theta)``.  Output: context ``[B, T+1, hidden]``.

The grid tokens are identical for every batch row, so they are built once ``[T+1, C]`` and the
batch axis is a stride-0 expand; conditioning is passed per batch row (see primitives/sit.py).
``rope_freqs`` stays a complex64 ``[2048, head_dim/2]`` buffer so the state-dict layout matches."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from ..config import EncoderConfig
from ..core.observations import grid_index
from ..primitives.embeddings import RotarySpec, SinusoidalEmbedding, precompute_freq_cis
from ..primitives.sit import SiT, SiTConfig


class ObservationContextEncoder(nn.Module):
    rope_freqs: Tensor

    def __init__(self, observation_dim: int, sde_param_dim: int, config: EncoderConfig) -> None:
        super().__init__()
        C = config.hidden_dim
        self.hidden_dim, self.num_heads = C, config.num_heads
        self.obs_proj = nn.Linear(observation_dim, C)
        self.bridge_token = nn.Parameter(torch.randn(C))
        self.time_embed = SinusoidalEmbedding(C)
        self.sde_param_proj = nn.Sequential(
            nn.Linear(sde_param_dim, config.cond_dim), nn.SiLU(),
            nn.Linear(config.cond_dim, config.cond_dim), nn.SiLU(),
            nn.Linear(config.cond_dim, config.cond_dim))
        self.register_buffer("rope_freqs", precompute_freq_cis(C // config.num_heads, end=2048))
        self.sit = SiT(SiTConfig(in_dim=C, hidden_dim=C, out_dim=C, cond_dim=config.cond_dim,
                                 num_heads=config.num_heads, depth=config.depth,
                                 mlp_hidden_dim=int(C * config.mlp_ratio)))

    def grid_tokens(self, obs_values: Tensor, obs_times: Tensor, time_horizon: float, time_step: float,
                    dtype: torch.dtype) -> Tensor:
        """``[T+1, C]`` tokens shared by all batch rows (reference: encoder.py:70-81)."""
        n = int(round(time_horizon / time_step)) + 1
        idx = grid_index(obs_times, time_step, n - 1)
        tokens = self.bridge_token.to(dtype).expand(n, -1).index_put((idx,), self.obs_proj(obs_values).to(dtype))
        return tokens + self._grid_embedding(n, float(time_horizon), obs_values.device, dtype)

    def _grid_embedding(self, n: int, time_horizon: float, device: torch.device, dtype: torch.dtype) -> Tensor:
        """Sinusoidal embedding of the time grid: a parameter-free function of (n, horizon), computed once per grid instead of nine
        small kernels per step."""
        key = (n, time_horizon, device, dtype, torch.is_autocast_enabled())
        cache = self.__dict__.setdefault("_grid_embed_cache", {})
        if key not in cache:
            grid = torch.linspace(0, time_horizon, n, device=device, dtype=dtype)
            value = self.time_embed(grid).to(dtype).detach()
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return value   # memory of a graph's private pool: valid for that graph only, never cached
            if len(cache) >= 8:
                cache.clear()
            cache[key] = value
        return cache[key]

    def _rotary(self, n: int, device: torch.device) -> RotarySpec:
        """Rotary tables of the first n positions (the spec caches its contiguous cos / sin copies: keep the spec across steps)."""
        key = (n, device)
        cache = self.__dict__.setdefault("_rotary_cache", {})
        if key not in cache:
            if len(cache) >= 8:
                cache.clear()
            freqs = self.rope_freqs
            if n > freqs.shape[0]:
                freqs = precompute_freq_cis(self.hidden_dim // self.num_heads, end=n, device=device)
            spec = RotarySpec.from_freqs(freqs[:n])
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return spec    # its cos / sin copies would live in the capturing graph's pool
            cache[key] = spec
        return cache[key]

    def forward(self, obs_values: Tensor, obs_times: Tensor, sde_parameters: Tensor, time_horizon: float,
                time_step: float) -> Tensor:
        B = sde_parameters.shape[0]
        tokens = self.grid_tokens(obs_values, obs_times, time_horizon, time_step, sde_parameters.dtype)
        n = tokens.shape[0]
        cond = self.sde_param_proj(sde_parameters)  # [B, cond], broadcast over tokens inside the blocks
        return self.sit(tokens.unsqueeze(0).expand(B, -1, -1), cond=cond, rotary=self._rotary(n, tokens.device), row_tokens=tokens)
