"""Sinusoidal time embedding and 1-D rotary embedding (reference: primitives/embeddings.py:10-83)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn


class SinusoidalEmbedding(nn.Module):
    """[..., dim] = cat(sin(t w_k), cos(t w_k)), w_k = max_period^(-k / (dim/2))."""

    def __init__(self, dim: int, max_period: float = 10000.0) -> None:
        super().__init__()
        if dim % 2:
            raise ValueError("dim must be even")
        self.dim, self.max_period = dim, max_period

    def forward(self, t: Tensor) -> Tensor:
        half = self.dim // 2
        k = torch.arange(half, device=t.device, dtype=t.dtype)
        w = torch.exp(-math.log(self.max_period) * k / half)
        phase = t.unsqueeze(-1) * w
        return torch.cat([phase.sin(), phase.cos()], dim=-1)


def precompute_freq_cis(dim: int, end: int = 1000, theta: float = 10000.0, *,
                        device: Optional[torch.device | str] = None,
                        dtype: Optional[torch.dtype] = None) -> Tensor:
    """complex64 table ``[end, dim/2]`` of exp(i * pos * theta^(-2k/dim))."""
    if dim % 2:
        raise ValueError("RoPE dimension must be even")
    dev = torch.device(device) if device is not None else None
    inv_freq = theta ** (-torch.arange(0, dim, 2, dtype=torch.float32, device=dev) / dim)
    pos = torch.arange(end, dtype=torch.float32, device=dev)
    ang = torch.outer(pos, inv_freq)
    table = torch.polar(torch.ones_like(ang), ang)
    if dtype == torch.float64:
        return table.to(torch.complex128)
    return table


@dataclass(frozen=True)
class RotarySpec:
    rotary_freqs: Tensor

    @classmethod
    def from_freqs(cls, freqs: Tensor) -> "RotarySpec":
        return cls(rotary_freqs=freqs)

    def cos_sin_tables(self, seq_len: int) -> tuple[Tensor, Tensor]:
        """Contiguous fp32 ``[seq_len, d/2]`` cos / sin tables (cached on the spec) for the fused kernels."""
        cache = self.__dict__.get("_tables")
        if cache is None or cache[0].shape[0] != seq_len:
            f = self.rotary_freqs
            if seq_len > f.shape[0]:
                raise ValueError("requested sequence length exceeds precomputed frequencies")
            f = f[:seq_len]
            cache = (f.real.float().contiguous(), f.imag.float().contiguous())
            object.__setattr__(self, "_tables", cache)
        return cache


def apply_rope_1d(x: Tensor, freqs: Tensor) -> Tensor:
    """Rotate ``x[..., seq, d]`` (half-split layout: first d/2 = real part, second = imaginary).

    Same map as multiplying ``complex(x[..., :h], x[..., h:])`` by ``freqs`` (what the reference
    does, embeddings.py:55-74), written with real arithmetic in the table's precision."""
    seq = x.shape[-2]
    if seq > freqs.shape[0]:
        raise ValueError("requested sequence length exceeds precomputed frequencies")
    rot = freqs.shape[-1] * 2
    half = rot // 2
    cos, sin = freqs[:seq].real.to(x.device), freqs[:seq].imag.to(x.device)
    re, im = x[..., :half].to(cos.dtype), x[..., half:rot].to(cos.dtype)
    out = torch.cat([re * cos - im * sin, re * sin + im * cos], dim=-1).to(x.dtype)
    if rot < x.shape[-1]:
        out = torch.cat([out, x[..., rot:]], dim=-1)
    return out
