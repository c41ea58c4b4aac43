"""SwiGLU feed-forward (reference: primitives/mlp.py:11-54): silu(a) * b with [a | b] = W_in x."""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from . import fused
from .initializer import init_linear_


_PAD = 128 if __import__("os").environ.get("VSDE_MLP_PAD") == "128" else 64   # VSDE_MLP_PAD=128: round-2 padding (A/B runs)


class SwiGLU(nn.Module):
    def __init__(self, in_dim: int, hidden_dim: int, *, bias: bool = True) -> None:
        super().__init__()
        self.in_dim, self.hidden_dim = in_dim, hidden_dim
        self.input_proj = init_linear_(nn.Linear(in_dim, 2 * hidden_dim, bias=bias))
        self.output_proj = init_linear_(nn.Linear(hidden_dim, in_dim, bias=bias))

    def _padded_weights(self, width: int):
        """Zero-pad the hidden width (682 at the example configs) to a multiple of 64 so that the GEMMs
        see 16-byte aligned rows / full MFMA tiles; padded units contribute silu(0)*0 = 0 exactly."""
        h, pad = self.hidden_dim, width - self.hidden_dim
        w1 = F.pad(self.input_proj.weight.view(2, h, self.in_dim), (0, 0, 0, pad)).reshape(2 * width, self.in_dim)
        b1 = None if self.input_proj.bias is None else F.pad(self.input_proj.bias.view(2, h), (0, pad)).reshape(2 * width)
        return w1, b1, F.pad(self.output_proj.weight, (0, pad))

    def padded_width(self) -> int:
        return -(-self.hidden_dim // _PAD) * _PAD   # 682 -> 704 (a multiple of the 64-column tile pairs; 768 wasted 9 % more)

    def packs(self, x: Tensor, fused_mlp: bool):
        """The (cached) bf16 operand packs of both projections, padded to ``padded_width()``."""
        width = self.padded_width()
        packs = getattr(self, "_packs", None)
        if (packs is None or packs[0].weight.device != x.device or packs[0].weight.shape[0] != 2 * width
                or (packs[0].grad_rows is not None) != fused_mlp):
            packs = fused.swiglu_packs(self.input_proj.weight, self.input_proj.bias, self.output_proj.weight,
                                       self.output_proj.bias, width, interleave=fused_mlp)
            object.__setattr__(self, "_packs", packs)
        return packs

    def forward(self, x: Tensor) -> Tensor:
        if fused.ENABLED and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not fused.fp16_autocast():
            width = self.padded_width()
            if fused.packed_linear_usable(x, 2 * width, self.in_dim):
                # bf16 operands of both projections (padded to `width`) live in a cache refreshed once per optimizer step
                fused_mlp = fused.swiglu_mlp_usable(x, width)   # both GEMMs with the SwiGLU math in their epilogues
                packs = self.packs(x, fused_mlp)
                if fused_mlp:
                    return fused.swiglu_mlp(x, packs[0], packs[1])
                return fused.packed_linear(fused.swiglu(fused.packed_linear(x, packs[0])), packs[1])
            if width != self.hidden_dim:
                w1, b1, w2 = self._padded_weights(width)
                return fused.linear(fused.swiglu(fused.linear(x, w1, b1)), w2, self.output_proj.bias)
            return fused.linear(fused.swiglu(fused.linear(x, self.input_proj.weight, self.input_proj.bias)),
                                self.output_proj.weight, self.output_proj.bias)
        a, b = self.input_proj(x).chunk(2, dim=-1)
        return self.output_proj(F.silu(a) * b)
