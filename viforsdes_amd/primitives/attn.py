"""Multi-head self attention of the SiT encoder (reference: primitives/attn.py:26-117):
QK RMS-norm (frozen unit weights, fp32), rotary embedding, optional value-residual mixing with the
first block's values, full (non-causal) SDPA, a sigmoid output gate shared across heads."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from contextlib import nullcontext

from . import fused
from .embeddings import RotarySpec, apply_rope_1d
from .initializer import init_linear_, zero_linear_
from .norm import RMS


def _sdpa_backend_context(t: Tensor):
    """On gfx950 / ROCm 7 the memory-efficient SDPA kernels are ~25 % faster (forward + backward, seq 401,
    head_dim 64, bf16) than the default flash path; prefer them, keep the others as fallbacks."""
    if not t.is_cuda:
        return nullcontext()
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        return sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True)
    except Exception:  # older torch: no priority API
        return nullcontext()


def _sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    with _sdpa_backend_context(q):
        return F.scaled_dot_product_attention(q, k, v, dropout_p=0.0)


class Attention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, *, qk_norm_eps: float = 1e-6, qk_norm: bool = True,
                 bias: bool = True, gate: bool = True, residual_v: bool = False) -> None:
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.qkv_proj = init_linear_(nn.Linear(embed_dim, 3 * embed_dim, bias=bias))
        self.out_proj = init_linear_(nn.Linear(embed_dim, embed_dim, bias=bias))
        self._use_gate = gate
        if gate:
            self.gate_proj = zero_linear_(nn.Linear(embed_dim, self.head_dim, bias=True))
        self._use_residual_v = residual_v
        if residual_v:
            self.v_residual_lambda = nn.Parameter(torch.tensor(0.5))
        make_norm = (lambda: RMS(self.head_dim, eps=qk_norm_eps, requires_grad=False)) if qk_norm else nn.Identity
        self.q_norm, self.k_norm = make_norm(), make_norm()

    def fusable(self, hidden_states: Tensor, rotary: Optional[RotarySpec]) -> bool:
        return (fused.ENABLED and rotary is not None and self._use_gate and isinstance(self.q_norm, RMS)
                and not self.q_norm.weight.requires_grad and not self.k_norm.weight.requires_grad
                and fused.usable(hidden_states, self.embed_dim, self.head_dim))

    def forward_fused(self, hidden_states: Tensor, *, rotary: RotarySpec, v0: Optional[Tensor],
                      v0link: Optional["fused.GradLink"] = None, defer_out: bool = False) -> tuple[Tensor, Tensor]:
        """Same map as ``forward(..., return_value=True)`` with the elementwise chains as fused HIP ops:
        qkv GEMM -> [RMS + RoPE + value mix + head layout] -> SDPA -> [sigmoid gate + merge heads] -> out GEMM.

        The per-head tensors live in memory as [B,N,h,d] (the layout the memory-efficient SDPA kernels read and
        write), handed around as [B,h,N,d] *views*, so neither direction needs a transposing copy."""
        N = hidden_states.shape[1]
        cos, sin = rotary.cos_sin_tables(N)
        mix = self._use_residual_v and v0 is not None
        lin = fused.linear
        # one projection for [q | k | v | gate logits]: one GEMM forward, one input-gradient GEMM and one
        # weight-gradient reduction backward, and the two consumers fill one gradient buffer between them
        n_out = self.qkv_proj.weight.shape[0] + self.gate_proj.weight.shape[0]
        if fused.packed_linear_usable(hidden_states, n_out, self.embed_dim):
            pack = getattr(self, "_proj_pack", None)
            if pack is None or pack.weight.device != hidden_states.device:
                has_b = self.qkv_proj.bias is not None
                pack = fused.row_pack([self.qkv_proj.weight, self.gate_proj.weight],
                                      [self.qkv_proj.bias, self.gate_proj.bias] if has_b else None)
                object.__setattr__(self, "_proj_pack", pack)
            if fused.attention_core_usable(hidden_states, pack, self.num_heads, self.head_dim, self.q_norm.weight, self.k_norm.weight, cos):
                # training step: projection epilogue = QK-norm / RoPE / value mix, attention store = gate + head merge, and the
                # attention backward's epilogues write the projection's gradient buffer (no raw [B,N,3C+d] tensor either way)
                out, v = fused.attention_core(hidden_states, pack, cos, sin, self.q_norm.weight, self.k_norm.weight,
                                              v0.transpose(1, 2) if mix else None, self.v_residual_lambda if mix else None,
                                              self.num_heads, self.q_norm.eps, self.head_dim ** -0.5, v0link,
                                              out_pack=fused.plain_pack(self.out_proj.weight, self.out_proj.bias))
                return out, v.transpose(1, 2)
            if fused.projection_split_nograd_usable(hidden_states, pack, self.num_heads, self.head_dim, self.q_norm.weight, cos):
                # posterior sampling: projection, QK-norm, RoPE, value mix and head layout in ONE kernel; no [B,N,3C+d] intermediate
                q, k, v, glog = fused.projection_split_nograd(hidden_states, pack, cos, sin, self.q_norm.weight, self.k_norm.weight,
                                                              v0.transpose(1, 2) if mix else None,
                                                              self.v_residual_lambda if mix else None, self.num_heads,
                                                              self.q_norm.eps)
                if fused.attention_usable(q):
                    out_tm = fused.attention(q, k, v, self.head_dim ** -0.5)
                else:
                    out_tm = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
                B_, N_ = hidden_states.shape[0], hidden_states.shape[1]
                if glog.shape[-1] == 64 and self.head_dim == 64 and self.out_proj.weight.shape[0] % 64 == 0:
                    # gate_merge folded into the out projection's operand load (one kernel instead of two)
                    po = fused.plain_pack(self.out_proj.weight, self.out_proj.bias)
                    if defer_out and fused.BLOCK_OUT_PROJ and self.out_proj.weight.shape[0] == self.embed_dim:
                        # the caller's block kernel applies gate, projection and bias in its prologue (csrc/vsde_mlp.hip, BLK == 2)
                        return (fused.DeferredOutProjection(out_tm.contiguous().view(B_, N_, self.embed_dim), glog.reshape(B_ * N_, 64), po),
                                v.transpose(1, 2))
                    wo, bo = po.operands()
                    y_out = fused._hip.linear_gated_bf16(out_tm.contiguous().view(B_ * N_, self.embed_dim), glog.reshape(B_ * N_, 64), wo, bo)
                    return y_out.view(B_, N_, -1), v.transpose(1, 2)
                merged = fused._hip.gate_merge_fwd(out_tm.contiguous(), glog, True)
                return lin(merged, self.out_proj.weight, self.out_proj.bias), v.transpose(1, 2)
            y = fused.packed_linear(hidden_states, pack)
        else:
            W = torch.cat([self.qkv_proj.weight, self.gate_proj.weight], dim=0)
            b = torch.cat([self.qkv_proj.bias, self.gate_proj.bias], dim=0) if self.qkv_proj.bias is not None else None
            y = lin(hidden_states, W, b)
        link = fused.GradLink()
        q, k, v = fused.attention_projection_split(y, cos, sin, self.q_norm.weight, self.k_norm.weight,
                                                   v0.transpose(1, 2) if mix else None,
                                                   self.v_residual_lambda if mix else None, self.num_heads,
                                                   self.q_norm.eps, True, link, v0link)
        if fused.attention_usable(q):
            out_tm = fused.attention(q, k, v, self.head_dim ** -0.5)  # token-major in, token-major out
        else:
            out_tm = _sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
        merged = fused.gate_merge_joint(out_tm, y, self.num_heads, True, link)
        return lin(merged, self.out_proj.weight, self.out_proj.bias), v.transpose(1, 2)  # values as a [B,h,N,d] view

    def forward(self, hidden_states: Tensor, *, rotary: Optional[RotarySpec] = None, v0: Optional[Tensor] = None,
                return_value: bool = False):
        B, N, _ = hidden_states.shape
        q, k, v = self.qkv_proj(hidden_states).view(B, N, 3, self.num_heads, self.head_dim).unbind(dim=2)
        q, k = self.q_norm(q), self.k_norm(k)
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)  # [B, h, N, d]
        if rotary is not None:
            q, k = apply_rope_1d(q, rotary.rotary_freqs), apply_rope_1d(k, rotary.rotary_freqs)
        if self._use_residual_v and v0 is not None:
            if v0.shape != v.shape:
                raise ValueError(f"v0 shape {tuple(v0.shape)} must match value heads {tuple(v.shape)}")
            lam = self.v_residual_lambda
            v = lam * v + (1.0 - lam) * v0
        out = _sdpa(q, k, v)
        if self._use_gate:
            out = out * torch.sigmoid(self.gate_proj(hidden_states)).unsqueeze(1)
        out = self.out_proj(out.transpose(1, 2).reshape(B, N, self.embed_dim))
        if self._use_residual_v or return_value:
            return out, v
        return out
