"""SiT transformer trunk of the observation encoder (reference: primitives/sit.py:18-186).

Per block: conditioning -> (scale, shift, gate) for the attention and MLP branches; each branch is
``x + gate * f((1 + scale) * LayerNorm(x) + shift)`` with a non-affine LayerNorm.  Blocks >= 1 mix
their values with block 0's values (value residual).

The conditioning tensor may be ``[B, cond]`` (one vector per batch row, broadcast over tokens) or
``[B, N, cond]``.  The reference always materialises the per-token form (models/encoder.py:85-86)
and runs the ``cond -> 6*dim`` linear per token; per-row evaluation is the same arithmetic N times
cheaper."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn

from . import fused
from .attn import Attention
from .cond import CondBranch, CondModulator
from .embeddings import RotarySpec
from .initializer import init_linear_
from .mlp import SwiGLU
from .norm import LayerNormConfig


@dataclass(frozen=True)
class SiTConfig:
    in_dim: int
    hidden_dim: int
    out_dim: int
    cond_dim: int
    num_heads: int
    depth: int
    mlp_hidden_dim: int
    bias: bool = True
    attn_gate: bool = True
    attn_residual_v: bool = True
    use_qk_norm: bool = True
    qk_norm_eps: float = 1e-6
    attn_norm: LayerNormConfig = LayerNormConfig(affine=False)
    mlp_norm: LayerNormConfig = LayerNormConfig(affine=False)


class SiTBlock(nn.Module):
    def __init__(self, *, dim: int, num_heads: int, mlp_hidden_dim: int, cond_dim: int, bias: bool = True,
                 attn_gate: bool = True, attn_residual_v: bool = False, use_qk_norm: bool = True,
                 qk_norm_eps: float = 1e-6, attn_norm: LayerNormConfig = LayerNormConfig(affine=False),
                 mlp_norm: LayerNormConfig = LayerNormConfig(affine=False)) -> None:
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self._cond_modulator = CondModulator(cond_dim=cond_dim, hidden_dim=dim, branches=2)
        self.self_attn = Attention(dim, num_heads, bias=bias, gate=attn_gate, qk_norm=use_qk_norm,
                                   qk_norm_eps=qk_norm_eps, residual_v=attn_residual_v)
        self.mlp = SwiGLU(dim, mlp_hidden_dim, bias=bias)
        self.attn_norm = attn_norm.build(dim=dim)
        self.mlp_norm = mlp_norm.build(dim=dim)

    def cond_params(self, *, cond: Tensor) -> tuple[CondBranch, ...]:
        return self._cond_modulator(cond=cond)

    def fusable(self, hidden_states: Tensor, cond: Tensor, rotary: Optional[RotarySpec]) -> bool:
        """Whether this block can take the fused HIP route (per-row conditioning, plain LayerNorms, gated QK-normed attention)."""
        plain = lambda n: isinstance(n, nn.LayerNorm) and not n.elementwise_affine
        return cond.ndim == 2 and plain(self.attn_norm) and plain(self.mlp_norm) and self.self_attn.fusable(hidden_states, rotary)

    def forward(self, hidden_states: Tensor, *, cond: Tensor, rotary: Optional[RotarySpec] = None,
                v0: Optional[Tensor] = None) -> tuple[Tensor, Tensor]:
        """Returns ``(hidden_states, value_heads)``; the caller keeps block 0's values as ``v0``."""
        if self.fusable(hidden_states, cond, rotary):
            # fused HIP route: 6 fused passes + 5 GEMMs + 1 attention call per block
            sa, ha, ga, sm, hm, gm = self._cond_modulator.net(cond).chunk(6, dim=-1)
            # each stream tensor feeds a norm and the residual that follows it: one GradLink per such pair
            link1, link2 = fused.GradLink(), fused.GradLink()
            h1 = fused.ln_modulate(hidden_states, sa, ha, self.attn_norm.eps, link1)
            attn_out, values = self.self_attn.forward_fused(h1, rotary=rotary, v0=v0)
            hidden_states = fused.gated_residual(hidden_states, attn_out, ga, link1)
            h2 = fused.ln_modulate(hidden_states, sm, hm, self.mlp_norm.eps, link2)
            return fused.gated_residual(hidden_states, self.mlp(h2), gm, link2), values
        if cond.ndim == 2:
            cond = cond.unsqueeze(1)  # broadcast over the token axis
        attn_mod, mlp_mod = self.cond_params(cond=cond)
        attn_out, values = self.self_attn(attn_mod.affine(self.attn_norm(hidden_states)), rotary=rotary, v0=v0,
                                          return_value=True)
        hidden_states = hidden_states + attn_mod.gate(attn_out)
        mlp_out = self.mlp(mlp_mod.affine(self.mlp_norm(hidden_states)))
        return hidden_states + mlp_mod.gate(mlp_out), values


class SiT(nn.Module):
    def __init__(self, config: SiTConfig) -> None:
        super().__init__()
        self.config = config
        self.blocks = nn.ModuleList([
            SiTBlock(dim=config.hidden_dim, num_heads=config.num_heads, mlp_hidden_dim=config.mlp_hidden_dim,
                     cond_dim=config.cond_dim, bias=config.bias, attn_gate=config.attn_gate,
                     attn_residual_v=config.attn_residual_v and idx > 0, use_qk_norm=config.use_qk_norm,
                     qk_norm_eps=config.qk_norm_eps, attn_norm=config.attn_norm, mlp_norm=config.mlp_norm)
            for idx in range(config.depth)])
        self.input_proj = init_linear_(nn.Linear(config.in_dim, config.hidden_dim, bias=config.bias))
        self.output_proj = init_linear_(nn.Linear(config.hidden_dim, config.out_dim, bias=config.bias))

    def _packed_modulations(self, cond: Tensor) -> Optional["fused.Modulations"]:
        """adaLN parameters of every block from ONE GEMM.  All blocks see the same conditioning vector, so their
        (SiLU -> Linear) modulators are one SiLU and one product against the row-stacked weights (cached bf16 operand); the
        result stays one [B, depth*6*C] tensor that the fused ops index by column range (``fused.Modulations``)."""
        nets = [blk._cond_modulator.net for blk in self.blocks]
        lin = [n[1] for n in nets]
        per = lin[0].weight.shape[0]
        C = self.config.hidden_dim
        if not (cond.is_cuda and cond.dtype == torch.bfloat16 and fused.ENABLED and all(isinstance(n[0], nn.SiLU) for n in nets)
                and all(l.bias is not None and l.weight.shape[0] == 6 * C for l in lin) and C % 8 == 0 and cond.shape[-1] % 8 == 0):
            return None
        pack = getattr(self, "_mods_pack", None)
        if pack is None or pack.weight.device != cond.device:
            pack = fused.row_pack([l.weight for l in lin], [l.bias for l in lin])
            object.__setattr__(self, "_mods_pack", pack)
        return fused.Modulations(fused.packed_linear(torch.nn.functional.silu(cond), pack), C)

    def _forward_fused_chain(self, tokens: Tensor, cond: Tensor, rotary: RotarySpec) -> Tensor:
        """All blocks on the fused route, with every gated residual fused into the LayerNorm that follows it -- the second
        norm of the same block, and the first norm of the NEXT block (each stream tensor is then written once and read once
        per direction): LN1 | attn | [res1+LN2] | mlp | [res2+next LN1] | attn | ..."""
        blocks = self.blocks
        nb = len(blocks)
        tokens = tokens.contiguous()  # a broadcast input is materialised once, not by each consumer
        mods = self._packed_modulations(cond)
        SA, HA, GA, SM, HM, GM = range(6)  # chunk order of a block's modulator output (primitives/sit.py:72)
        # the input stream feeds the first norm and the first residual: their two gradient contributions meet inside the norm's
        # backward kernel (link0) instead of in a separate accumulation pass
        link0 = fused.GradLink()
        if mods is None:
            ml = [blk._cond_modulator.net(cond).chunk(6, dim=-1) for blk in blocks]
            h1 = fused.ln_modulate(tokens, ml[0][SA], ml[0][HA], blocks[0].attn_norm.eps, link0)
        else:
            h1 = fused.ln_modulate_m(tokens, mods, (0, SA), (0, HA), blocks[0].attn_norm.eps, final=True, link=link0)
        v0: Optional[Tensor] = None
        v0link = fused.GradLink() if self.config.attn_residual_v else None  # one buffer for the value-residual gradient
        for k, blk in enumerate(blocks):
            block_form = (isinstance(blk.mlp, SwiGLU) and blk.mlp.input_proj.bias is not None
                          and fused.mlp_block_nograd_usable(tokens, mods, blk.mlp.padded_width())
                          and fused.swiglu_mlp_usable(tokens, blk.mlp.padded_width()))
            attn_out, values = blk.self_attn.forward_fused(h1, rotary=rotary, v0=v0, v0link=v0link, defer_out=block_form)
            if v0 is None and self.config.attn_residual_v:
                v0 = values
            lk = link0 if k == 0 else None
            if block_form:
                # no-grad call: [res1 + LN2 | MLP | res2 + next LN1] is ONE kernel (csrc/vsde_mlp.hip, block form)
                pin, pout = blk.mlp.packs(tokens, True)
                tokens, h1 = fused.mlp_block_nograd(tokens, attn_out, mods, k, k + 1 if k + 1 < nb else None, blk.mlp_norm.eps,
                                                    blocks[k + 1].attn_norm.eps if k + 1 < nb else blk.mlp_norm.eps, pin, pout)
                continue
            if mods is None:
                x1, h2 = fused.residual_norm(tokens, attn_out, ml[k][GA], ml[k][SM], ml[k][HM], blk.mlp_norm.eps, lk)
            else:
                x1, h2 = fused.residual_norm_m(tokens, attn_out, mods, (k, GA), (k, SM), (k, HM), blk.mlp_norm.eps, lk)
            mlp_out = blk.mlp(h2)
            if k + 1 < nb:
                eps = blocks[k + 1].attn_norm.eps
                if mods is None:
                    tokens, h1 = fused.residual_norm(x1, mlp_out, ml[k][GM], ml[k + 1][SA], ml[k + 1][HA], eps)
                else:
                    tokens, h1 = fused.residual_norm_m(x1, mlp_out, mods, (k, GM), (k + 1, SA), (k + 1, HA), eps)
            else:
                tokens = (fused.gated_residual(x1, mlp_out, ml[k][GM]) if mods is None
                          else fused.gated_residual_m(x1, mlp_out, mods, (k, GM)))
        return tokens

    def forward(self, x: Tensor, *, cond: Tensor, rotary: Optional[RotarySpec] = None, row_tokens: Optional[Tensor] = None) -> Tensor:
        """``row_tokens`` (additive): the [N, in_dim] sequence that ``x`` broadcasts over the batch, when the caller has it --
        ``x[0]`` of the expanded view costs a full-size zero fill + a batch sum in the backward (SelectBackward / ExpandBackward)."""
        if x.ndim == 3 and x.stride(0) == 0 and x.shape[0] > 1:
            # the same token sequence for every batch row (the observation grid): project it once, then broadcast
            row = row_tokens if (row_tokens is not None and row_tokens.shape == x.shape[1:]) else x[0]
            tokens = self.input_proj(row).unsqueeze(0).expand(x.shape[0], -1, -1)
        else:
            tokens = self.input_proj(x)
        if all(block.fusable(tokens, cond, rotary) for block in self.blocks):
            return fused.linear(self._forward_fused_chain(tokens, cond, rotary), self.output_proj.weight, self.output_proj.bias)
        v0: Optional[Tensor] = None
        for block in self.blocks:
            tokens, values = block(tokens, cond=cond, rotary=rotary, v0=v0)
            if v0 is None and self.config.attn_residual_v:
                v0 = values
        return fused.linear(tokens, self.output_proj.weight, self.output_proj.bias)
