"""Autograd wrappers of the fused HIP encoder operators (csrc/vsde_encoder.hip).

Each function is one HBM pass forward and one backward; they are numerically the chains of
primitives/{sit,attn,mlp}.py (which remain the eager specification and the CPU path of the
encoder).  ``usable(x)`` decides whether a tensor can take the fused route."""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from .. import _hip

ENABLED = True  # set False to force the unfused torch chains (A/B tests)


def fp16_autocast() -> bool:
    """fp16 autocast (``TrainingConfig(amp_dtype=AmpDtype.FLOAT16)``, reference config.py:24-38): the fused encoder operators are
    bf16 / fp32 kernels, so the whole encoder then runs as the torch autocast chain (library GEMMs, SDPA); the GRU head takes the
    fp16 context as fp32."""
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16


def usable(x: Tensor, channels: int, head_dim: int) -> bool:
    half = head_dim // 2
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not fp16_autocast() and channels % 64 == 0 and channels <= 1024
            and head_dim % 2 == 0 and 1 <= half <= 64 and (half & (half - 1)) == 0)


class GradLink:
    """One-shot mailbox between two autograd Functions that consume the SAME tensor in one forward pass.

    Autograd would add their two gradient contributions with a separate full-size kernel.  Instead the Function whose
    backward is guaranteed to run first (it sits downstream of the other one in the graph) parks its contribution here
    and reports ``None``; the upstream Function's backward kernel folds it into the gradient it writes anyway.  Create
    one link per tensor per forward call and hand it to exactly that pair (see ``SiTBlock._forward_fused``)."""
    __slots__ = ("value",)

    def __init__(self) -> None:
        self.value: Optional[Tensor] = None

    def take(self) -> Optional[Tensor]:
        v, self.value = self.value, None
        return v


class _LnModulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, eps, link):
        x, scale, shift = x.contiguous(), scale.to(x.dtype).contiguous(), shift.to(x.dtype).contiguous()
        y, mean, rstd = _hip.ln_modulate_fwd(x, scale, shift, eps)
        ctx.save_for_backward(x, scale, mean, rstd)
        ctx.link = link
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, scale, mean, rstd = ctx.saved_tensors
        dres = ctx.link.take() if ctx.link is not None else None  # gradient reaching x through the residual branch
        dx, dscale, dshift = _hip.ln_modulate_bwd(x, scale, dy.to(x.dtype).contiguous(), mean, rstd, dres)
        return dx, dscale, dshift, None, None


def ln_modulate(x: Tensor, scale: Tensor, shift: Tensor, eps: float = 1e-5, link: Optional[GradLink] = None) -> Tensor:
    """``LayerNorm(x) * (1 + scale[:, None]) + shift[:, None]`` for x [B,N,C], scale/shift [B,C].

    ``link``: shared with the ``gated_residual`` that consumes the same ``x`` downstream of this op's output."""
    return _LnModulate.apply(x, scale, shift, eps, link)


class _GatedResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, gate, link):
        x, y, gate = x.contiguous(), y.to(x.dtype).contiguous(), gate.to(x.dtype).contiguous()
        ctx.save_for_backward(y, gate)
        ctx.link = link
        return _hip.gated_residual_fwd(x, y, gate)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        y, gate = ctx.saved_tensors
        dout = dout.to(y.dtype).contiguous()
        dy, dgate = _hip.gated_residual_bwd(y, gate, dout)
        if ctx.link is not None and ctx.needs_input_grad[0]:
            ctx.link.value = dout  # picked up by the ln_modulate backward of the same x
            return None, dy, dgate, None
        return dout, dy, dgate, None


def gated_residual(x: Tensor, y: Tensor, gate: Tensor, link: Optional[GradLink] = None) -> Tensor:
    """``x + gate[:, None] * y``.  ``link``: see ``ln_modulate`` (y must depend on that ln_modulate's output)."""
    return _GatedResidual.apply(x, y, gate, link)


class _ResidualNorm(torch.autograd.Function):
    """xnew = x + gate*y; h = LayerNorm(xnew)*(1+scale)+shift in one pass; the backward folds the gradient reaching xnew from
    its other consumers, the norm backward, gate*dx and the three token sums into one kernel."""

    @staticmethod
    def forward(ctx, x, y, gate, scale, shift, eps, link=None):
        x = x.contiguous()
        y, gate, scale, shift = (t.to(x.dtype).contiguous() for t in (y, gate, scale, shift))
        xnew, h, mean, rstd = _hip.residual_ln_fwd(x, y, gate, scale, shift, eps)
        ctx.save_for_backward(xnew, y, gate, scale, mean, rstd)
        ctx.link = link
        return xnew, h

    @staticmethod
    @once_differentiable
    def backward(ctx, dxnew, dh):
        xnew, y, gate, scale, mean, rstd = ctx.saved_tensors
        dh = torch.zeros_like(xnew) if dh is None else dh.to(xnew.dtype).contiguous()
        dxnew = None if dxnew is None else dxnew.to(xnew.dtype).contiguous()
        dx, dy, dgate, dscale, dshift = _hip.residual_ln_bwd(xnew, y, gate, scale, dh, dxnew, mean, rstd)
        if ctx.link is not None and ctx.needs_input_grad[0]:
            ctx.link.value, dx = dx, None   # picked up by the ln_modulate backward of the same x (no separate accumulation pass)
        return dx, dy, dgate, dscale, dshift, None, None


def residual_norm(x: Tensor, y: Tensor, gate: Tensor, scale: Tensor, shift: Tensor, eps: float = 1e-5,
                  link: Optional[GradLink] = None) -> tuple[Tensor, Tensor]:
    """``xnew = x + gate[:, None]*y`` and ``h = LayerNorm(xnew)*(1+scale[:, None]) + shift[:, None]`` -> (xnew, h).
    ``link``: shared with the ``ln_modulate`` that consumed the same ``x`` upstream of ``y`` (see ``ln_modulate``)."""
    return _ResidualNorm.apply(x, y, gate, scale, shift, eps, link)


class Modulations:
    """The adaLN parameters of ALL blocks as one [B, depth*6*C] tensor (output of a single GEMM) plus the gradient buffer
    that the consuming ops fill in place.

    The ``*_m`` ops below read their scale / shift / gate straight out of ``allm`` (row pitch = its width: no chunk copies)
    and their backward passes write dscale / dshift / dgate straight into ``grad``; the op that ran FIRST in the forward
    pass (``final=True``: its backward runs last, everything downstream depends on it) hands ``grad`` to autograd as the
    gradient of ``allm``.  Every chunk is consumed by exactly one op, so the buffer is complete at that point."""

    def __init__(self, allm: Tensor, channels: int) -> None:
        self.allm, self.C, self.grad = allm, channels, None

    def vec(self, block: int, j: int) -> Tensor:
        o = (6 * block + j) * self.C
        return self.allm.detach()[:, o:o + self.C]

    def gvec(self, block: int, j: int) -> Tensor:
        if self.grad is None:
            self.grad = torch.zeros_like(self.allm)
        o = (6 * block + j) * self.C
        return self.grad[:, o:o + self.C]

    def take_grad(self) -> Tensor:
        g, self.grad = (self.grad if self.grad is not None else torch.zeros_like(self.allm)), None
        return g


class _LnModulateM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, allm, mods, sc, sh, eps, final, link=None):
        x = x.contiguous()
        y, mean, rstd = _hip.ln_modulate_fwd(x, mods.vec(*sc), mods.vec(*sh), eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.meta = (mods, sc, sh, final, link)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        mods, sc, sh, final, link = ctx.meta
        dres = link.take() if link is not None else None   # gradient reaching x through the residual branch
        dx, _, _ = _hip.ln_modulate_bwd(x, mods.vec(*sc), dy.to(x.dtype).contiguous(), mean, rstd, dres,
                                        dscale=mods.gvec(*sc), dshift=mods.gvec(*sh))
        return dx, (mods.take_grad() if final else None), None, None, None, None, None, None


class _ResidualNormM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, allm, mods, gt, sc, sh, eps, link=None):
        x = x.contiguous(); y = y.to(x.dtype).contiguous()
        xnew, h, mean, rstd = _hip.residual_ln_fwd(x, y, mods.vec(*gt), mods.vec(*sc), mods.vec(*sh), eps)
        ctx.save_for_backward(xnew, y, mean, rstd)
        ctx.meta = (mods, gt, sc, sh, link)
        return xnew, h

    @staticmethod
    @once_differentiable
    def backward(ctx, dxnew, dh):
        xnew, y, mean, rstd = ctx.saved_tensors
        mods, gt, sc, sh, link = ctx.meta
        dh = torch.zeros_like(xnew) if dh is None else dh.to(xnew.dtype).contiguous()
        dxnew = None if dxnew is None else dxnew.to(xnew.dtype).contiguous()
        dx, dy, _, _, _ = _hip.residual_ln_bwd(xnew, y, mods.vec(*gt), mods.vec(*sc), dh, dxnew, mean, rstd,
                                               dgate=mods.gvec(*gt), dscale=mods.gvec(*sc), dshift=mods.gvec(*sh))
        if link is not None and ctx.needs_input_grad[0]:
            link.value, dx = dx, None   # picked up by the ln_modulate backward of the same x
        return dx, dy, None, None, None, None, None, None, None


class _GatedResidualM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, allm, mods, gt):
        x = x.contiguous(); y = y.to(x.dtype).contiguous()
        ctx.save_for_backward(y)
        ctx.meta = (mods, gt)
        return _hip.gated_residual_fwd(x, y, mods.vec(*gt))

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        mods, gt = ctx.meta
        dout = dout.to(y.dtype).contiguous()
        dy, _ = _hip.gated_residual_bwd(y, mods.vec(*gt), dout, dgate=mods.gvec(*gt))
        return dout, dy, None, None, None


def ln_modulate_m(x: Tensor, mods: Modulations, scale: tuple[int, int], shift: tuple[int, int], eps: float, final: bool = False,
                  link: Optional[GradLink] = None) -> Tensor:
    """``ln_modulate`` with scale / shift = chunks ``(block, j)`` of ``mods`` (see ``Modulations``)."""
    return _LnModulateM.apply(x, mods.allm, mods, scale, shift, eps, final, link)


def residual_norm_m(x: Tensor, y: Tensor, mods: Modulations, gate: tuple[int, int], scale: tuple[int, int], shift: tuple[int, int],
                    eps: float, link: Optional[GradLink] = None) -> tuple[Tensor, Tensor]:
    return _ResidualNormM.apply(x, y, mods.allm, mods, gate, scale, shift, eps, link)


def gated_residual_m(x: Tensor, y: Tensor, mods: Modulations, gate: tuple[int, int]) -> Tensor:
    return _GatedResidualM.apply(x, y, mods.allm, mods, gate)


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        u = u.contiguous()
        ctx.save_for_backward(u)
        return _hip.swiglu_fwd(u)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (u,) = ctx.saved_tensors
        return _hip.swiglu_bwd(u, dout.to(u.dtype).contiguous())


def swiglu(u: Tensor) -> Tensor:
    """``silu(u[..., :h]) * u[..., h:]``."""
    return _SwiGLU.apply(u)


class _GateMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, glog, token_major):
        attn, glog = attn.contiguous(), glog.to(attn.dtype).contiguous()
        ctx.save_for_backward(attn, glog)
        ctx.token_major = token_major
        return _hip.gate_merge_fwd(attn, glog, token_major)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        attn, glog = ctx.saved_tensors
        dattn, dglog = _hip.gate_merge_bwd(attn, glog, dout.to(attn.dtype).contiguous(), ctx.token_major)
        return dattn, dglog, None


def gate_merge(attn: Tensor, gate_logits: Tensor, token_major: bool = False) -> Tensor:
    """[B,h,N,d] (or [B,N,h,d] when ``token_major``) x sigmoid([B,N,d]) -> [B,N,h*d]."""
    return _GateMerge.apply(attn, gate_logits, token_major)


class _GateMergeJoint(torch.autograd.Function):
    """gate_merge reading its logits from columns [off, off+d) of the merged [qkv | gate] projection ``y``.  Its
    backward allocates the gradient buffer of ``y``, fills the gate columns and parks it in ``link`` for
    ``_QkNormRopeJoint.backward`` (always later: q, k, v feed the attention whose output this op consumes)."""

    @staticmethod
    def forward(ctx, attn, y, off, token_major, link):
        attn = attn.contiguous()
        ctx.save_for_backward(attn, y)
        ctx.meta = (off, token_major, link)
        d = attn.shape[-1]
        return _hip.gate_merge_fwd(attn, y[..., off:off + d], token_major)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        attn, y = ctx.saved_tensors
        off, token_major, link = ctx.meta
        d = attn.shape[-1]
        dy = torch.empty_like(y) if off + d == y.shape[-1] else torch.zeros_like(y)
        dattn, _ = _hip.gate_merge_bwd(attn, y[..., off:off + d], dout.to(attn.dtype).contiguous(), token_major,
                                       dglog=dy[..., off:off + d])
        link.value = dy
        return dattn, None, None, None, None


class _QkNormRopeJoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, cos, sin, wq, wk, v0, lam, heads, eps, token_major, width, link, v0link):
        v0c = v0.to(y.dtype).contiguous() if v0 is not None else None
        lamc = lam.detach().float().reshape(1).contiguous() if lam is not None else None
        q, k, v = _hip.qk_norm_rope_fwd(y[..., :width], cos, sin, wq, wk, v0c, lamc, heads, eps, token_major)
        ctx.save_for_backward(y, cos, sin, wq, wk, v0c, lamc)
        ctx.meta = (heads, eps, lam.dtype if lam is not None else None, token_major, width, link, v0link)
        return q, k, v

    @staticmethod
    @once_differentiable
    def backward(ctx, dq, dk, dv):
        y, cos, sin, wq, wk, v0, lam = ctx.saved_tensors
        heads, eps, lam_dtype, token_major, width, link, v0link = ctx.meta
        dy = link.take()
        if dy is None:  # the gate branch did not take part in this backward pass
            dy = torch.zeros_like(y)
        c = lambda t: t.to(y.dtype).contiguous()
        # value-residual gradient: every consumer block adds its share into ONE buffer (v0link) inside its kernel and reports
        # None; the block that produced v0 (v0 is None here) runs last and folds the buffer into its own dv
        acc = v0link.value if (v0link is not None and v0 is not None) else None
        extra = v0link.take() if (v0link is not None and v0 is None) else None
        _, dv0, dlam = _hip.qk_norm_rope_bwd(y[..., :width], cos, sin, wq, wk, v0, lam, c(dq), c(dk), c(dv), heads, eps,
                                             token_major, dqkv=dy[..., :width], dv0=acc, dv_extra=extra)
        if v0link is not None and v0 is not None:
            v0link.value, dv0 = dv0, None
        if dlam is not None:
            dlam = dlam.to(lam_dtype).reshape(())
        return dy, None, None, None, None, dv0, dlam, None, None, None, None, None, None


def attention_projection_split(y: Tensor, cos: Tensor, sin: Tensor, wq: Tensor, wk: Tensor, v0: Optional[Tensor],
                               lam: Optional[Tensor], heads: int, eps: float, token_major: bool, link: GradLink,
                               v0link: Optional[GradLink] = None):
    """(q, k, v) from the first 3C columns of the merged [qkv | gate] projection ``y`` [B,N,3C+d].  ``v0link``: shared by
    the block that produces the residual values (called with ``v0=None``) and every block that mixes them in."""
    d = y.shape[-1] // (3 * heads + 1)
    return _QkNormRopeJoint.apply(y.contiguous(), cos, sin, wq, wk, v0, lam, heads, eps, token_major, 3 * heads * d, link, v0link)


def projection_split_nograd_usable(x: Tensor, pack: "PackedWeight", heads: int, d: int, wq: Tensor, cos: Tensor) -> bool:
    """The no-grad form of [qkv | gate] projection + attention_projection_split as ONE kernel (GEMM with the QK-norm / RoPE /
    value-mix epilogue): K = 256, head_dim 64, a gate block that is a multiple of 64 wide, fp32 norm weights and tables."""
    rows = x.numel() // x.shape[-1]
    return (OWN_GEMM and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] in (128, 256) and d == 64
            and pack.weight.shape[0] >= 3 * heads * 64 and (pack.weight.shape[0] - 3 * heads * 64) % 64 == 0 and rows > 0
            and wq.dtype == torch.float32 and cos.dtype == torch.float32 and cos.shape[-1] == 32)


@torch.no_grad()
def projection_split_nograd(x: Tensor, pack: "PackedWeight", cos: Tensor, sin: Tensor, wq: Tensor, wk: Tensor,
                            v0: Optional[Tensor], lam: Optional[Tensor], heads: int, eps: float):
    """(q, k, v [B,N,heads,64], gate logits [B,N,G]) for x [B,N,256]; v0 token-major [B,N,heads,64] or None."""
    B, N, K = x.shape
    w, b = pack.operands()
    v0c = v0.to(torch.bfloat16).contiguous() if v0 is not None else None
    lamc = lam.detach().float().reshape(1).contiguous() if lam is not None and v0 is not None else None
    q, k, v, g = _hip.linear_qknorm_bf16(x.reshape(B * N, K), w, b, heads, N, cos.contiguous(), sin.contiguous(), wq.contiguous(),
                                         wk.contiguous(), v0c, lamc, eps)
    shape = (B, N, heads, 64)
    return q.view(shape), k.view(shape), v.view(shape), (g.view(B, N, -1) if g is not None else None)


def gate_merge_joint(attn: Tensor, y: Tensor, heads: int, token_major: bool, link: GradLink) -> Tensor:
    """gate_merge with the gate logits taken from the last d columns of ``y`` (see attention_projection_split)."""
    d = y.shape[-1] // (3 * heads + 1)
    return _GateMergeJoint.apply(attn, y.contiguous(), 3 * heads * d, token_major, link)


class _QkNormRope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cos, sin, wq, wk, v0, lam, heads, eps, token_major):
        qkv = qkv.contiguous()
        v0c = v0.to(qkv.dtype).contiguous() if v0 is not None else None
        lamc = lam.detach().float().reshape(1).contiguous() if lam is not None else None
        q, k, v = _hip.qk_norm_rope_fwd(qkv, cos, sin, wq, wk, v0c, lamc, heads, eps, token_major)
        ctx.save_for_backward(qkv, cos, sin, wq, wk, v0c, lamc)
        ctx.meta = (heads, eps, lam.dtype if lam is not None else None, token_major)
        return q, k, v

    @staticmethod
    @once_differentiable
    def backward(ctx, dq, dk, dv):
        qkv, cos, sin, wq, wk, v0, lam = ctx.saved_tensors
        heads, eps, lam_dtype, token_major = ctx.meta
        c = lambda t: t.to(qkv.dtype).contiguous()
        dqkv, dv0, dlam = _hip.qk_norm_rope_bwd(qkv, cos, sin, wq, wk, v0, lam, c(dq), c(dk), c(dv), heads, eps, token_major)
        if dlam is not None:
            dlam = dlam.to(lam_dtype).reshape(())
        return dqkv, None, None, None, None, dv0, dlam, None, None, None


def qk_norm_rope(qkv: Tensor, cos: Tensor, sin: Tensor, wq: Tensor, wk: Tensor, v0: Optional[Tensor], lam: Optional[Tensor],
                 heads: int, eps: float, token_major: bool = False) -> tuple[Tensor, Tensor, Tensor]:
    """qkv [B,N,3C] -> (q, k, v) each [B,heads,N,d] ([B,N,heads,d] when ``token_major``; v0 in the same layout):
    RMS-norm + RoPE on q,k; v = lam*v + (1-lam)*v0."""
    return _QkNormRope.apply(qkv, cos, sin, wq, wk, v0, lam, heads, eps, token_major)


class _Attention(torch.autograd.Function):
    """softmax(scale q k^T) v on token-major heads [B,N,H,64] (csrc/vsde_attn.hip): forward with K and V of one head resident
    in LDS; backward as two deterministic kernels (dq | dk, dv) that recompute the probabilities from the saved log-sum-exp."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o, lse = _hip.attention_fwd(q, k, v, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = _hip.attention_bwd(do.to(q.dtype).contiguous(), q, k, v, o, lse, ctx.scale)
        return dq, dk, dv, None


def attention_usable(q: Tensor) -> bool:
    """q token-major [B,N,H,d]: bf16, head_dim 64 or 128 (any sequence length: short head_dim-64 sequences run the LDS-resident
    kernels, everything else the kernels that stream K / V tiles)."""
    return ENABLED and q.is_cuda and q.dtype == torch.bfloat16 and q.ndim == 4 and q.shape[-1] in (64, 128)


def attention(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """Token-major [B,N,H,64] self-attention core (no mask, no dropout); see ``attention_usable``."""
    return _Attention.apply(q, k, v, scale)


# Parameters change behind ``Tensor._version``'s back in two common ways: torch's own fused AdamW / Adam kernels do not bump it
# (measured on torch 2.10: ``AdamW(fused=True).step()`` leaves ``p._version`` untouched) and neither does the trainer's own
# optimizer kernel.  Every optimizer step therefore also advances this epoch (a global ``Optimizer.step`` post-hook, and
# inference/fused_optimizer.py calls ``note_parameters_changed`` itself): a pack is fresh only if versions AND epoch match.
_param_epoch = 0


def note_parameters_changed() -> None:
    global _param_epoch
    _param_epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_post_hook  # noqa: E402



def _on_optimizer_step(opt, args, kwargs) -> None:
    """Only optimizers that own a parameter of a live pack advance the epoch (an unrelated optimizer elsewhere in the process
    used to mark every pack stale: one needless refresh per step of that other model)."""
    ids = getattr(opt, "_vsde_param_ids", None)
    n = sum(len(g["params"]) for g in opt.param_groups)
    if ids is None or ids[0] != n:
        ids = (n, {id(p) for g in opt.param_groups for p in g["params"]})
        try:
            opt._vsde_param_ids = ids
        except Exception:
            pass
    if any(id(q) in ids[1] for pk in PackedWeight._live for q in pk.params):
        note_parameters_changed()


_register_step_post_hook(_on_optimizer_step)


class PackedWeight:
    """bf16 GEMM operand (weight [rows, cols] and optional bias [rows]) assembled from fp32 parameter row blocks, kept across
    optimizer steps.

    The encoder's Linears need their weights in bf16 (autocast), some of them concatenated ([qkv | gate]) or zero-padded
    (SwiGLU width 682 -> 768).  With torch ops that is a cat/pad plus a cast per operand per step and their autograd nodes.
    A pack owns the assembled bf16 tensors, re-fills them only when a source parameter changed (``Tensor._version``), and
    hands the fp32 gradient of the packed operand back to the parameters as views.  ``refresh_all(force=True)`` (called by the
    trainer right after the optimizer step) re-fills every live pack with one multi-tensor copy.

    ``weight_pieces`` / ``bias_pieces``: lists of ``(param, src_row, n, dst_row)``: rows ``src_row:src_row+n`` of the 2-D
    (1-D) parameter land in rows ``dst_row:dst_row+n`` of the packed weight (bias), columns ``:param.shape[1]``; everything
    not covered stays zero."""

    _live = weakref.WeakSet()

    def __init__(self, rows: int, cols: int, weight_pieces, bias_pieces, device, grad_rows: Optional[Tensor] = None) -> None:
        self.weight = torch.zeros(rows, cols, device=device, dtype=torch.bfloat16)
        self.bias = torch.zeros(rows, device=device, dtype=torch.bfloat16) if bias_pieces else None
        self.weight_pieces, self.bias_pieces = list(weight_pieces), list(bias_pieces or [])
        self.weight_t: Optional[Tensor] = None   # [cols, rows] copy for the input-gradient GEMM, kept in step by refresh_all()
        self._t_versions: Optional[list[int]] = None
        # packs whose rows are a permutation of ONE parameter (interleaved SwiGLU halves): packed row of every parameter row
        self.grad_rows = grad_rows
        # its inverse for the weight-gradient kernel: output (parameter) row of every packed row, -1 for padding rows
        self.grad_row_map: Optional[Tensor] = None
        if grad_rows is not None:
            self.grad_row_map = torch.full((rows,), -1, device=device, dtype=torch.int32)
            self.grad_row_map[grad_rows] = torch.arange(grad_rows.numel(), device=device, dtype=torch.int32)
        self.params: list[Tensor] = []
        for p, *_ in self.weight_pieces + self.bias_pieces:
            if not any(p is q for q in self.params):
                self.params.append(p)
        self._versions: Optional[list[int]] = None
        PackedWeight._live.add(self)

    def _copy_lists(self):
        dst, src = [], []
        for p, s0, n, d0 in self.weight_pieces:
            dst.append(self.weight[d0:d0 + n, :p.shape[1]]); src.append(p.detach()[s0:s0 + n])
        for p, s0, n, d0 in self.bias_pieces:
            dst.append(self.bias[d0:d0 + n]); src.append(p.detach()[s0:s0 + n])
        return dst, src

    def stale(self) -> bool:
        return self._versions != [_param_epoch] + [p._version for p in self.params]

    def mark_fresh(self) -> None:
        self._versions = [_param_epoch] + [p._version for p in self.params]

    @torch.no_grad()
    def operands(self):
        if self.stale():
            for d, s_ in zip(*self._copy_lists()):
                d.copy_(s_)
            self.mark_fresh()
        return self.weight, self.bias

    @torch.no_grad()
    def transposed(self) -> Tensor:
        """The packed weight as [cols, rows] (what ``dx = dy W`` needs as an ``x W^T`` GEMM operand)."""
        self.operands()
        if self.weight_t is None:
            self.weight_t = torch.empty(self.weight.shape[1], self.weight.shape[0], device=self.weight.device, dtype=torch.bfloat16)
        if self._t_versions != self._versions:
            self.weight_t.copy_(self.weight.t())
            self._t_versions = list(self._versions)
        return self.weight_t

    def split_grads(self, dW: Tensor, db: Optional[Tensor], mapped: bool = False) -> list[Optional[Tensor]]:
        """fp32 gradient of the packed operand -> one gradient per entry of ``self.params`` (a view when the parameter is
        one whole row block, otherwise its row blocks concatenated; a row gather for permuted packs)."""
        if self.grad_rows is not None:   # params = [weight] or [weight, bias], rows permuted
            if mapped:   # the kernel already stored the rows in parameter order (grad_row_map)
                return [dW if q.ndim == 2 else db for q in self.params]
            return [dW.index_select(0, self.grad_rows) if q.ndim == 2 else db.index_select(0, self.grad_rows) for q in self.params]
        out: list[Optional[Tensor]] = []
        for q in self.params:
            blocks = [dW[d0:d0 + n, :p.shape[1]] for p, s0, n, d0 in sorted(self.weight_pieces, key=lambda t: t[1]) if p is q]
            if not blocks and db is not None:
                blocks = [db[d0:d0 + n] for p, s0, n, d0 in sorted(self.bias_pieces, key=lambda t: t[1]) if p is q]
            out.append(None if not blocks else blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=0))
        return out

    _tables: dict = {}      # key (device, pack identities, buffer addresses) -> device table of refresh tiles
    _captured_tables: list = []   # tables a HIP-graph capture has seen: a replay launches with the table it was captured with,
                                  # so these are never freed (~90 KB each)

    def _refresh_tiles(self) -> list[tuple]:
        """(src, dst, dst_t, src_pitch, dst_pitch, pitch_t, rows, cols) per tile of <= 16 rows: csrc/vsde_pack.hip."""
        out = []
        wt = self.weight_t
        for p, s0, n, d0 in self.weight_pieces:
            if p.dtype != torch.float32 or p.stride(1) != 1:
                return []
            for r0 in range(0, n, 16):
                out.append((p.data_ptr() + 4 * (s0 + r0) * p.stride(0), self.weight.data_ptr() + 2 * (d0 + r0) * self.weight.stride(0),
                            0 if wt is None else wt.data_ptr() + 2 * (d0 + r0), p.stride(0), self.weight.stride(0),
                            0 if wt is None else wt.stride(0), min(16, n - r0), p.shape[1]))
        for p, s0, n, d0 in self.bias_pieces:
            if p.dtype != torch.float32:
                return []
            out.append((p.data_ptr() + 4 * s0, self.bias.data_ptr() + 2 * d0, 0, n, n, 0, 1, n))
        return out

    @staticmethod
    @torch.no_grad()
    def refresh_all(force: bool = False, params: Optional[set] = None) -> None:
        """Re-fill every live pack whose sources changed.  ``force``: re-fill all of them regardless of the version counters --
        what the trainer does right after the optimizer step: the fused (multi-tensor) AdamW kernel updates the parameters
        WITHOUT bumping ``Tensor._version``, so a version check alone would keep the first step's operands forever.
        ``params`` (ids of parameters): only the packs built from these -- a trainer passes its own model's, so that its step
        (and a HIP graph captured from it, which replays with raw addresses) never touches the packs of another model that
        happens to be alive; those stay covered by their staleness check.
        On the GPU a forced refresh is ONE kernel over a cached table of tiles (fp32 parameters -> bf16 packs and their
        transposes, csrc/vsde_pack.hip); otherwise one multi-tensor copy.  The table is built (a host -> device copy) on first
        use; during a stream capture, where that copy is illegal, a missing table means the multi-tensor copy is captured instead."""
        live = sorted(PackedWeight._live, key=id)
        if params is not None:
            live = [pk for pk in live if any(id(q) in params for q in pk.params)]
        if force and live and all(pk.weight.is_cuda for pk in live):
            by_dev: dict = {}
            for pk in live:
                by_dev.setdefault(pk.weight.device, []).append(pk)
            ok = True
            for dev, packs in by_dev.items():
                key = (dev, tuple((id(pk), pk.weight.data_ptr(), 0 if pk.weight_t is None else pk.weight_t.data_ptr(),
                                   tuple(q.data_ptr() for q in pk.params)) for pk in packs))
                table = PackedWeight._tables.get(key)
                if table is None and torch.cuda.is_current_stream_capturing():
                    ok = False
                    break
                if table is None:
                    rows = []
                    for pk in packs:
                        tiles = pk._refresh_tiles()
                        if not tiles:
                            ok = False
                            break
                        rows += [(a, b, c, d, e, f, r | (cl << 32), 0) for a, b, c, d, e, f, r, cl in tiles]
                    if not ok:
                        break
                    if len(PackedWeight._tables) > 64:   # a long-lived process that keeps building models: drop old tables
                        PackedWeight._tables.clear()     # (captured ones stay referenced from _captured_tables)
                    table = PackedWeight._tables[key] = torch.tensor(rows, dtype=torch.int64).to(dev)
                if torch.cuda.is_current_stream_capturing() and not any(t is table for t in PackedWeight._captured_tables):
                    PackedWeight._captured_tables.append(table)
                _hip.pack_refresh(table)
            if ok:
                for pk in live:
                    pk.mark_fresh()
                    if pk.weight_t is not None:
                        pk._t_versions = list(pk._versions)
                PackedWeight._refresh_derived(live, force=True)
                return
        dst, src, packs = [], [], []
        for pk in live:
            if force or pk.stale():
                d, s_ = pk._copy_lists()
                dst += d; src += s_; packs.append(pk)
        if dst:
            torch._foreach_copy_(dst, src)
            for pk in packs:
                pk.mark_fresh()
                if pk.weight_t is not None:
                    pk.weight_t.copy_(pk.weight.t())
                    pk._t_versions = list(pk._versions)
        PackedWeight._refresh_derived(live, force=force)

    # operands derived from packs (tile images of the fused MLP kernels): objects with ``packs``, ``_key`` (None = dirty) and
    # ``refresh_if_stale()``.  Inside a stream capture they are rebuilt with their packs, so that a captured training step (whose
    # forward never checks staleness) replays with images of the parameters its captured optimizer step has just written.  Eagerly
    # they are only MARKED dirty and rebuilt by the next ``operands()`` / ``refresh_dirty_derived()`` -- the training step never
    # reads them, and five small launches per image and step are not free on a launch-bound step.
    _derived = weakref.WeakSet()

    @staticmethod
    def _refresh_derived(live, force: bool = False) -> None:
        ids = {id(pk) for pk in live}
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        for d in sorted(PackedWeight._derived, key=id):
            if not any(id(pk) in ids for pk in d.packs):
                continue
            if force:
                d._key = None      # the host-side version counters say nothing after a forced refresh (fused optimizer, graph replay)
            if capturing:
                d.refresh_if_stale()

    @staticmethod
    def invalidate_derived(params: Optional[set] = None) -> None:
        """Mark the derived images (of the packs built from ``params``) dirty: their packs were rewritten on the device behind the
        host counters' back (a replayed HIP graph of the training step)."""
        for d in PackedWeight._derived:
            if params is None or any(id(q) in params for pk in d.packs for q in pk.params):
                d._key = None

    @staticmethod
    @torch.no_grad()
    def refresh_dirty_derived(params: Optional[set] = None) -> None:
        """Rebuild every dirty derived image now (what a captured sampling call needs before its replay: its graph reads the
        images in place and never calls ``operands()``)."""
        for d in sorted(PackedWeight._derived, key=id):
            if params is None or any(id(q) in params for pk in d.packs for q in pk.params):
                d.refresh_if_stale()


# the MFMA GEMM kernels of csrc/vsde_linear.hip for the shapes they cover (VSDE_OWN_GEMM=0: library GEMMs everywhere, for A/B runs)
OWN_GEMM = os.environ.get("VSDE_OWN_GEMM", "1") != "0"
# also the older non-persistent deep-reduction (cols) variant for shapes the persistent deep kernel does not take
OWN_GEMM_COLS = os.environ.get("VSDE_OWN_GEMM_COLS", "0") == "1"


# Rows below which the encoder Linears keep the library GEMM (the own kernels are laid out for stripes of 128 / 256 rows on 256
# CUs: a few thousand rows do not fill them).  VSDE_OWN_GEMM_MIN_ROWS changes it; the first call below it says so once.
OWN_GEMM_MIN_ROWS = int(os.environ.get("VSDE_OWN_GEMM_MIN_ROWS", "4096"))
_small_m_logged = False


def own_gemm(M: int, N: int, K: int, epilogue: int = 0) -> bool:
    """Whether y[M,N] = x[M,K] W[N,K]^T runs on the own MFMA kernels."""
    global _small_m_logged
    if not (ENABLED and OWN_GEMM):
        return False
    if M < OWN_GEMM_MIN_ROWS:
        if not _small_m_logged:
            _small_m_logged = True
            import logging
            logging.getLogger("viforsdes_amd").info(
                "encoder Linear with %d rows (< %d): hipBLASLt instead of the own MFMA kernels (VSDE_OWN_GEMM_MIN_ROWS)", M, OWN_GEMM_MIN_ROWS)
        return False
    variant = _hip.linear_variant(M, N, K, epilogue)
    return variant in (1, 3) or (variant == 2 and OWN_GEMM_COLS)


def _mm_nt(x2: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    """x2 [M,K] @ w[N,K]^T (+ bias) in bf16: own kernel when the shape is covered, hipBLASLt otherwise."""
    if x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and own_gemm(x2.shape[0], w.shape[0], w.shape[1]):
        return _hip.linear_bf16(x2, w, bias)
    return torch.nn.functional.linear(x2, w, bias)


# ---- weight gradients of the packed Linears: immediately, or collected over one backward pass -----------------------------------
# At a few thousand rows (the OU example: 12.9 k tokens) every weight gradient is a ~20 us kernel + a ~10 us reduction that cannot
# fill the chip alone, 25 of each per step.  Inside ``deferred_weight_grads()`` (the trainer wraps its backward pass in it) the
# backward functions below queue (dy, x) instead, return no gradient for the pack's parameters, and the context's exit issues all of
# them in a handful of launches (``_hip.linear_wgrad_group``: problem by problem the same arithmetic, bit-identical) and accumulates
# into ``p.grad`` the way autograd would have.  Large problems (>= WGRAD_DEFER_MAX_ROWS rows) fill the chip by themselves and read
# their dy while it is still warm in the cache: they stay immediate.
# Contract of the deferred path: the gradients are written to ``p.grad`` directly at the context's exit, NOT through autograd's
# AccumulateGrad -- post-accumulate-grad hooks (and a DDP reducer, if someone wraps the model) do not fire for these parameters;
# inference/data_parallel.py packs them after the backward (its early bucket only ever holds hook-announced gradients).  Only
# leaves that require a gradient are deferred; several parameters of one pack may end up with ``.grad`` views of one buffer.
# The queue keeps its (dy, x) operands alive until the flush: it is flushed early whenever it holds more than
# WGRAD_DEFER_MAX_BYTES (VSDE_WGRAD_DEFER_MAX_BYTES, default 1 GiB).
WGRAD_DEFER_MAX_ROWS = int(os.environ.get("VSDE_WGRAD_DEFER_MAX_ROWS", "65536"))   # 0: never defer
WGRAD_DEFER_MAX_BYTES = int(os.environ.get("VSDE_WGRAD_DEFER_MAX_BYTES", str(1 << 30)))
_wgrad_queue: Optional[list] = None
_wgrad_queue_bytes = 0
_deferred_ever: set = set()      # ids of every parameter a deferred product has been queued for (inference/data_parallel.py keeps
                                 # them out of its early bucket: their gradient is complete only when the queue is flushed)


def deferred_parameter_ids(pending_only: bool = False) -> set:
    """ids of the parameters written by the deferred path; ``pending_only``: those with a product in the open queue right now."""
    if not pending_only:
        return set(_deferred_ever)
    if not _wgrad_queue:
        return set()
    return {id(q) for item in _wgrad_queue for q in item[5].params}


def _queue_weight_grads(item: tuple) -> None:
    """Append one (dy, x, ...) problem to the open queue; flush the queue when its operands exceed the byte budget."""
    global _wgrad_queue_bytes
    _wgrad_queue.append(item)
    _deferred_ever.update(id(q) for q in item[5].params)
    _wgrad_queue_bytes += item[0].numel() * item[0].element_size() + item[1].numel() * item[1].element_size()
    if _wgrad_queue_bytes > WGRAD_DEFER_MAX_BYTES:
        queue = list(_wgrad_queue)
        _wgrad_queue.clear()
        _wgrad_queue_bytes = 0
        _flush_weight_grads(queue)


class deferred_weight_grads:
    def __enter__(self):
        global _wgrad_queue, _wgrad_queue_bytes
        self.outer, _wgrad_queue = _wgrad_queue, []
        self.outer_bytes, _wgrad_queue_bytes = _wgrad_queue_bytes, 0
        return self

    def __exit__(self, exc_type, exc, tb):
        global _wgrad_queue, _wgrad_queue_bytes
        queue, _wgrad_queue = _wgrad_queue, self.outer
        _wgrad_queue_bytes = self.outer_bytes
        if exc_type is None and queue:
            _flush_weight_grads(queue)
        return False


@torch.no_grad()
def _flush_weight_grads(queue: list) -> None:
    results = _hip.linear_wgrad_group([q[:5] for q in queue], group_plan=True)
    for (dy, x, want_bias, row_map, out_rows, pack, mapped), (dW, db) in zip(queue, results):
        for prm, g in zip(pack.params, pack.split_grads(dW, db, mapped=mapped)):
            if g is None or not prm.requires_grad:
                continue
            if g.dtype != prm.dtype:
                g = g.to(prm.dtype)
            if prm.grad is None:
                prm.grad = g if g.is_contiguous() else g.contiguous()   # (autograd copies a strided gradient the same way)
            else:
                prm.grad.add_(g)


def _pack_weight_grads(dy: Tensor, x: Tensor, pack: "PackedWeight", row_map: Optional[Tensor] = None, out_rows: Optional[int] = None) -> list:
    """Gradients of ``pack.params`` for y = x W^T + b with W = the pack: a list for the backward's return value (all ``None`` when
    the product was queued for the end of the backward pass)."""
    want_bias, mapped = pack.bias is not None, row_map is not None
    if (_wgrad_queue is not None and 0 < dy.shape[0] < WGRAD_DEFER_MAX_ROWS
            and all(q.is_leaf and q.requires_grad for q in pack.params)):   # gradients can only be written to ``.grad`` of leaves
        _queue_weight_grads((dy, x, want_bias, row_map, out_rows, pack, mapped))
        return [None] * len(pack.params)
    dW, db = _hip.linear_wgrad(dy, x, want_bias, row_map, out_rows)
    return pack.split_grads(dW, db, mapped=mapped)


class _LinearParams:
    """What ``_flush_weight_grads`` needs of a plain (unpacked) Linear: its parameters and how the product maps to them."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor]):
        self.params = [weight] if bias is None else [weight, bias]
        self.bias = bias

    def leaf(self) -> bool:   # gradients can only be written to ``.grad`` of leaves: anything else takes the immediate path
        return all(q.is_leaf and q.requires_grad for q in self.params)

    def split_grads(self, dW: Tensor, db: Optional[Tensor], mapped: bool = False) -> list:
        return [dW] if self.bias is None else [dW, db]


class _PackedLinear(torch.autograd.Function):
    """y = x W^T + b with a ``PackedWeight``: forward and input gradient on the MFMA GEMM kernels of csrc/vsde_linear.hip
    where the shape is covered (hipBLASLt otherwise), HIP weight-gradient kernel, gradients returned per parameter piece."""

    @staticmethod
    def forward(ctx, x, pack, *params):
        wb, bb = pack.operands()
        x2 = x.reshape(-1, x.shape[-1])
        y = _mm_nt(x2, wb, bb).reshape(*x.shape[:-1], wb.shape[0])
        ctx.save_for_backward(x, wb)
        ctx.pack = pack
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        pack = ctx.pack
        dy2 = dy.to(torch.bfloat16).reshape(-1, dy.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if own_gemm(dy2.shape[0], wb.shape[1], wb.shape[0]):
            dx = _hip.linear_bf16(dy2, pack.transposed(), None).reshape(x.shape)   # dx = dy W as an x W^T product with W^T
        else:
            dx = (dy2 @ wb).reshape(x.shape)
        return (dx, None, *_pack_weight_grads(dy2, x2, pack))


def packed_linear_usable(x: Tensor, rows: int, cols: int) -> bool:
    n = x.numel() // x.shape[-1]
    return ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and n >= OWN_GEMM_MIN_ROWS and rows % 8 == 0 and cols % 8 == 0


def packed_linear(x: Tensor, pack: PackedWeight) -> Tensor:
    return _PackedLinear.apply(x, pack, *pack.params)


class _Linear(torch.autograd.Function):
    """y = x W^T + b for bf16 activations: forward and the input gradient stay on hipBLASLt, the weight/bias
    gradient (a reduction over ~2e5 rows that hipBLASLt runs at a few % of the HBM roofline) is one fused HIP
    kernel pair (csrc/vsde_wgrad.hip) with fp32, deterministic results."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        wb = weight.to(torch.bfloat16)
        y = torch.nn.functional.linear(x, wb, None if bias is None else bias.to(torch.bfloat16))
        ctx.save_for_backward(x, wb)
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype)
        ctx.owner = _LinearParams(weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        wdtype, bdtype = ctx.meta
        dy2 = dy.to(torch.bfloat16).reshape(-1, dy.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        dx = (dy2 @ wb).reshape(x.shape)
        own = ctx.owner
        if (_wgrad_queue is not None and 0 < dy2.shape[0] < WGRAD_DEFER_MAX_ROWS and own.leaf()):
            _queue_weight_grads((dy2, x2, bdtype is not None, None, None, own, False))   # issued with the others at the end of the pass
            return dx, None, None
        dW, db = _hip.linear_wgrad(dy2, x2, bdtype is not None)
        return dx, dW.to(wdtype), None if db is None else db.to(bdtype)


_PLAIN_PACKS: dict[int, tuple] = {}  # id(weight) -> (weakref to the weight, pack); tensors cannot be dict keys (== is elementwise)


def plain_pack(weight: Tensor, bias: Optional[Tensor]) -> PackedWeight:
    """The (cached) pack of an ordinary Linear: one weight, one optional bias."""
    ent = _PLAIN_PACKS.get(id(weight))
    pk = ent[1] if ent is not None and ent[0]() is weight else None
    if pk is None or pk.weight.device != weight.device or (pk.bias is None) != (bias is None):
        n = weight.shape[0]
        pk = PackedWeight(n, weight.shape[1], [(weight, 0, n, 0)], None if bias is None else [(bias, 0, n, 0)], weight.device)
        key = id(weight)
        _PLAIN_PACKS[key] = (weakref.ref(weight, lambda _r, k=key, d=_PLAIN_PACKS: d.pop(k, None)), pk)
    return pk


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """Drop-in for ``F.linear`` inside the encoder: large bf16 inputs go through the packed bf16 operand cache and the HIP
    weight-gradient kernel."""
    if (isinstance(weight, torch.nn.Parameter) and (bias is None or isinstance(bias, torch.nn.Parameter))
            and packed_linear_usable(x, weight.shape[0], weight.shape[1])):
        return packed_linear(x, plain_pack(weight, bias))
    rows = x.numel() // x.shape[-1]
    if (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and rows >= OWN_GEMM_MIN_ROWS and weight.shape[0] % 8 == 0
            and weight.shape[1] % 8 == 0 and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)):
        return _Linear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


# VSDE_FUSED_MLP=0: the two-launch SwiGLU MLP everywhere (A/B runs)
FUSED_MLP = os.environ.get("VSDE_FUSED_MLP", "1") != "0"
# VSDE_BLOCK_MLP=0: residual / LayerNorm passes stay separate kernels in the no-grad chain (A/B runs)
BLOCK_MLP = os.environ.get("VSDE_BLOCK_MLP", "1") != "0"
# block form: also the attention branch's gate + out projection in the kernel's prologue (VSDE_BLOCK_OUT_PROJ=0: separate launch)
BLOCK_OUT_PROJ = os.environ.get("VSDE_BLOCK_OUT_PROJ", "1") != "0"
# VSDE_FUSED_MLP_BWD=1: the training step's MLP backward as one kernel (du + dx) instead of rows kernel + library GEMM (opt-in, see above)
FUSED_MLP_BWD = os.environ.get("VSDE_FUSED_MLP_BWD", "0") == "1"
# one 256-row workgroup per CU and no column chunks: below ~128 rows per CU the two-launch form (column chunks fill the chip) is faster
BLOCK_MLP_MIN_ROWS = int(os.environ.get("VSDE_BLOCK_MLP_MIN_ROWS", "32768"))


class MlpImages:
    """Weight operands of the fused SwiGLU MLP kernels (csrc/vsde_mlp.hip) as tile images, one tile = 16 hidden units:
    ``w1`` [T, 32 rows, C + 8] (row 8 g + 4 h + i = (a if g < 2 else b) unit 16 t + 8 h + 4 (g & 1) + i: the MFMA result layout then
    hands every lane a_j, b_j for the 8 units it feeds into the next product), ``w2`` [T, 2, C, 8] (W_out[n, 16 t + 8 h + 0..7]) and
    ``b1`` [T, 64] fp32 (b_in in w1's row order).  Built from the bf16 packs ``pin`` ([2 width, C]; halves interleaved in blocks of
    16 rows, or at rows 0 / width) and ``pout`` ([C, width]) and rebuilt whenever those are refreshed."""

    def __init__(self, pin: PackedWeight, pout: PackedWeight, width: int) -> None:
        C, T, dev = pin.weight.shape[1], width // 16, pin.weight.device
        w1b, w2b, b1b = _hip.mlp_image_bytes(C)
        self.w1 = torch.zeros(T, w1b // 2, device=dev, dtype=torch.bfloat16)
        self.w2 = torch.zeros(T, w2b // 2, device=dev, dtype=torch.bfloat16)
        self.b1 = torch.zeros(T, b1b // 4, device=dev, dtype=torch.float32)
        rho = torch.arange(32)
        g, h, i = rho >> 3, (rho >> 2) & 1, rho & 3
        ab, j = g >> 1, 8 * h + 4 * (g & 1) + i
        t = torch.arange(T)[:, None]
        rows = (32 * t + 16 * ab + j) if pin.grad_rows is not None else (ab * width + 16 * t + j)
        self.rows = rows.reshape(-1).to(dev)
        self.packs, self.width, self.C, self.T = (pin, pout), width, C, T
        self._key = None
        PackedWeight._derived.add(self)

    @torch.no_grad()
    def refresh_if_stale(self) -> None:
        pin, pout = self.packs
        key = (tuple(pin._versions or ()), tuple(pout._versions or ()))
        if key == self._key:
            return
        C, T = self.C, self.T
        self.w1[:, :32 * (C + 8)].view(T, 32, C + 8)[:, :, :C].copy_(pin.weight.index_select(0, self.rows).view(T, 32, C))
        if pin.bias is not None:
            self.b1[:, :32].copy_(pin.bias.index_select(0, self.rows).view(T, 32))
        self.w2.view(T, 2, C, 8).copy_(pout.weight.view(C, T, 2, 8).permute(1, 2, 0, 3))
        self._key = key

    def operands(self):
        for pk in self.packs:
            pk.operands()
        self.refresh_if_stale()
        return self.w1, self.w2, self.b1


class OutProjImage:
    """The attention out projection's weight [C, C] as C / 16 tiles in the W2 image format ([2 h][C][8]: W_o[n, 16 t + 8 h + 0..7]),
    the operand of the block kernel's out-projection prologue (csrc/vsde_mlp.hip, BLK == 2); rebuilt whenever the pack is refreshed."""

    def __init__(self, pack: PackedWeight) -> None:
        C = pack.weight.shape[0]
        self.img = torch.zeros(C // 16, 2, C, 8, device=pack.weight.device, dtype=torch.bfloat16)
        self.packs, self.C, self._key = (pack,), C, None   # `.packs`: the protocol of PackedWeight._derived
        PackedWeight._derived.add(self)

    @torch.no_grad()
    def refresh_if_stale(self) -> None:
        pack = self.packs[0]
        key = tuple(pack._versions or ())
        if key == self._key:
            return
        C = self.C
        self.img.copy_(pack.weight.view(C, C // 16, 2, 8).permute(1, 2, 0, 3))
        self._key = key

    def operand(self) -> Tensor:
        self.packs[0].operands()
        self.refresh_if_stale()
        return self.img


class DeepImage:
    """A [256, K] weight (or, ``transposed``, the [K, 256] weight whose transpose is meant) as K / 16 k-step images [2 h][256][8]
    (W[n, 16 t + 8 h + 0..7]): the operand of ``_hip.linear_deep256`` (csrc/vsde_mlp.hip::deep256_kernel), rebuilt whenever the pack
    is refreshed."""

    def __init__(self, pack: PackedWeight, transposed: bool) -> None:
        K = pack.weight.shape[0] if transposed else pack.weight.shape[1]
        self.img = torch.zeros(K // 16, 2, 256, 8, device=pack.weight.device, dtype=torch.bfloat16)
        self.packs, self.K, self.transposed, self._key = (pack,), K, transposed, None
        PackedWeight._derived.add(self)

    @torch.no_grad()
    def refresh_if_stale(self) -> None:
        pack = self.packs[0]
        key = tuple(pack._versions or ())
        if key == self._key:
            return
        K = self.K
        if self.transposed:   # weight [K, 256]: W^T[n, k] = weight[k, n]
            self.img.copy_(pack.weight.view(K // 16, 2, 8, 256).permute(0, 1, 3, 2))
        else:                 # weight [256, K]
            self.img.copy_(pack.weight.view(256, K // 16, 2, 8).permute(1, 2, 0, 3))
        self._key = key

    def operand(self) -> Tensor:
        self.packs[0].operands()
        self.refresh_if_stale()
        return self.img


# the deep-reduction GEMMs at width 256 (SwiGLU output projection, the two input-gradient products of a block) on the own kernels
# csrc/vsde_mlp.hip::deep256q_kernel / deep256p_kernel.  OPT-IN (VSDE_DEEP256=1) and only with the tools' library (VSDE_HIP_LIB =
# libvsde_hip_abl.so): correct, and 10-15 % slower than the library's 256 x 224 macro tiles (112 | 193 | 126 us against 96 | 167 | 113 us
# for K = 704 | 1408 | 832 at the LV shape; three schedules measured, profiles/r06_deep256_ablation.txt)
DEEP256 = os.environ.get("VSDE_DEEP256", "0") == "1"


def deep256_usable(M: int, N: int, K: int) -> bool:
    return (ENABLED and OWN_GEMM and DEEP256 and N == 256 and K % 64 == 0 and K >= 512 and M >= BLOCK_MLP_MIN_ROWS
            and _hip.has_ablations())   # the kernels exist only in the tools' library (csrc/vsde_common.h)


def deep256(x2: Tensor, pack: PackedWeight, transposed: bool, bias: Optional[Tensor]) -> Tensor:
    """x2 [M, K] @ W^T (+ bias) with W = pack.weight [256, K] (or its transpose when the pack holds [K, 256])."""
    attr = "_deep_image_t" if transposed else "_deep_image"
    img = getattr(pack, attr, None)
    if img is None:
        img = DeepImage(pack, transposed)
        setattr(pack, attr, img)
    return _hip.linear_deep256(x2, img.operand(), bias)


class DeferredOutProjection:
    """What ``SelfAttention.forward_fused(defer_out=True)`` hands back instead of the projected branch output in no-grad calls:
    the merged attention output [B, N, C], the gate logits [B*N, 64] and the out projection's pack -- the block kernel applies
    gate, projection and bias in its prologue (``mlp_block_nograd``)."""

    def __init__(self, attn: Tensor, glog: Tensor, pack: PackedWeight) -> None:
        self.attn, self.glog, self.pack = attn, glog, pack


class MlpBwdImages:
    """Weight operand of the fused SwiGLU MLP backward (csrc/vsde_mlp.hip::mlp_bwd_kernel): per pair tile P (32 hidden units) one
    contiguous image [W2T | W1] --
    ``W2T`` [32 units, C + 8] (+ padding to whole KB): W_out[:, 32 P + tau], the A operand of ds = dy W_out;
    ``W1`` [4 k-steps, 2 h, C, 8]: k-step 2 ab + q, slot 4 g' + i holds the W_in row of (ab, unit 32 P + 8 (2 q + g') + 4 h + i), so that the
    du values a lane computes are, in register order, its B fragments of dx += du W_in.
    Built from the interleaved bf16 packs (``swiglu_packs(..., interleave=True)``) and rebuilt whenever those are refreshed."""

    def __init__(self, pin: PackedWeight, pout: PackedWeight, width: int) -> None:
        assert pin.grad_rows is not None, "the fused backward works on the 16-row interleaved u / du layout"
        C, TP, dev = pin.weight.shape[1], width // 32, pin.weight.device
        self.nbytes = _hip.mlp_bwd_image_bytes(C)
        self.w2_elems = 32 * (C + 8)
        w2_bytes = (32 * (2 * C + 16) + 1023) // 1024 * 1024
        self.w1_off = w2_bytes // 2
        self.img = torch.zeros(TP, self.nbytes // 2, device=dev, dtype=torch.bfloat16)
        ks, h, e = torch.meshgrid(torch.arange(4), torch.arange(2), torch.arange(8), indexing="ij")
        ab, q, gp, i = ks >> 1, ks & 1, e >> 2, e & 3
        unit = 8 * (2 * q + gp) + 4 * h + i                      # within the pair tile
        rows = 32 * (unit // 16) + 16 * ab + unit % 16            # row within the tile's 64 interleaved rows
        self.rows = (64 * torch.arange(TP)[:, None] + rows.reshape(1, -1)).reshape(-1).to(dev)
        self.packs, self.width, self.C, self.TP = (pin, pout), width, C, TP
        self._key = None
        PackedWeight._derived.add(self)

    @torch.no_grad()
    def refresh_if_stale(self) -> None:
        pin, pout = self.packs
        key = (tuple(pin._versions or ()), tuple(pout._versions or ()))
        if key == self._key:
            return
        C, TP = self.C, self.TP
        self.img[:, :self.w2_elems].view(TP, 32, C + 8)[:, :, :C].copy_(pout.weight.t().reshape(TP, 32, C))
        w1 = pin.weight.index_select(0, self.rows).view(TP, 4, 2, 8, C).permute(0, 1, 2, 4, 3)
        self.img[:, self.w1_off:self.w1_off + 4 * 2 * C * 8].view(TP, 4, 2, C, 8).copy_(w1)
        self._key = key

    def operand(self) -> Tensor:
        for pk in self.packs:
            pk.operands()
        self.refresh_if_stale()
        return self.img


def mlp_block_nograd_usable(x: Tensor, mods: Optional["Modulations"], width: int) -> bool:
    """Whether [gated residual + modulated LayerNorm + SwiGLU MLP + gated residual + next LayerNorm] runs as ONE kernel
    (``mlp_block_nograd``): no-grad calls on bf16 [B, N, C] streams with C in (128, 256) and packed modulations."""
    return (ENABLED and OWN_GEMM and FUSED_MLP and BLOCK_MLP and not torch.is_grad_enabled() and mods is not None and x.is_cuda
            and x.dtype == torch.bfloat16 and x.ndim == 3 and x.shape[-1] in (128, 256) and width % 64 == 0 and width >= 64
            and x.shape[1] >= 86 and x.numel() // x.shape[-1] >= BLOCK_MLP_MIN_ROWS and mods.allm.dtype == torch.bfloat16)


@torch.no_grad()
def mlp_block_nograd(x: Tensor, attn_out, mods: "Modulations", block: int, nxt: Optional[int], eps: float, eps_next: float,
                     pin: PackedWeight, pout: PackedWeight):
    """(tokens_new, h_next) of one SiT block's second half (reference primitives/sit.py:112-128) from the stream ``x`` and the
    attention branch's output (a tensor, or a ``DeferredOutProjection``: the kernel then runs the out projection too):
    csrc/vsde_mlp.hip in its block form.  ``nxt``: index of the block whose first norm follows (None:
    last block, ``h_next`` is None).  Chunk order of a block's modulations: (sa, ha, ga, sm, hm, gm)."""
    img = getattr(pin, "_mlp_images", None)
    if img is None:
        img = pin._mlp_images = MlpImages(pin, pout, pout.weight.shape[1])
    w1i, w2i, b1i = img.operands()
    sn = None if nxt is None else mods.vec(nxt, 0)
    hs = None if nxt is None else mods.vec(nxt, 1)
    if isinstance(attn_out, DeferredOutProjection):
        po = attn_out.pack
        oimg = getattr(po, "_out_image", None)
        if oimg is None:
            oimg = po._out_image = OutProjImage(po)
        return _hip.mlp_attn_block_fwd(x.contiguous(), attn_out.attn, attn_out.glog, oimg.operand(), po.bias, mods.vec(block, 2),
                                       mods.vec(block, 3), mods.vec(block, 4), mods.vec(block, 5), sn, hs, eps, eps_next, w1i, w2i, b1i,
                                       pout.bias, pout.weight.shape[1])
    return _hip.mlp_block_fwd(x.contiguous(), attn_out.to(x.dtype).contiguous(), mods.vec(block, 2), mods.vec(block, 3), mods.vec(block, 4),
                              mods.vec(block, 5), sn, hs, eps, eps_next, w1i, w2i, b1i, pout.bias, pout.weight.shape[1])


class _SwiGLUMLP(torch.autograd.Function):
    """SwiGLU feed-forward ``W_out (silu(a) * b) + b_out`` with ``[a | b] = W_in x + b_in`` (mlp.py:50-54) around two GEMM
    kernels with fused epilogues: the input projection writes u (for the backward) and s = silu(a) * b in one pass, the
    backward computes ds = dy W_out inside the kernel that turns it into du = swiglu'(u) ds.  ``pin`` packs W_in with its
    halves interleaved in blocks of 16 rows (``swiglu_packs(..., interleave=True)``)."""

    @staticmethod
    def forward(ctx, x, pin, pout, train, *params):
        w1, b1 = pin.operands()
        w2, b2 = pout.operands()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if not train and FUSED_MLP and x2.shape[1] in (128, 256) and w2.shape[1] % 64 == 0 and x2.shape[0] >= BLOCK_MLP_MIN_ROWS:
            # no-grad call (posterior sampling): ONE kernel, neither u nor s is materialised (csrc/vsde_mlp.hip)
            img = getattr(pin, "_mlp_images", None)
            if img is None:
                img = pin._mlp_images = MlpImages(pin, pout, w2.shape[1])
            w1i, w2i, b1i = img.operands()
            y, _ = _hip.mlp_fwd(x2, w1i, w2i, b1i, b2, w2.shape[1])
            return y.reshape(*x.shape[:-1], w2.shape[0])
        # the pre-activation u is only kept for the backward: a no-grad call (posterior sampling) skips its [M, 2*width] write
        u, s_ = _hip.linear_swiglu_bf16(x2, w1, b1, want_u=train)
        y = deep256(s_, pout, False, b2) if deep256_usable(s_.shape[0], w2.shape[0], w2.shape[1]) else _mm_nt(s_, w2, b2)
        if train:
            ctx.save_for_backward(x2, u, s_, w1)
        ctx.packs = (pin, pout)
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, u, s_, w1 = ctx.saved_tensors
        pin, pout = ctx.packs
        dy2 = dy.to(torch.bfloat16).reshape(-1, dy.shape[-1]).contiguous()
        if (FUSED_MLP_BWD and pin.grad_rows is not None and dy2.shape[1] in (128, 256) and pout.weight.shape[1] % 64 == 0
                and dy2.shape[0] >= BLOCK_MLP_MIN_ROWS and _hip.has_ablations()):
            # one kernel for du AND dx (csrc/vsde_mlp.hip::mlp_bwd_kernel): du is never re-read.  Opt-in: at the LV shape it measures
            # 470-480 us against 435 us for the two launches below (profiles/r05_mlp_bwd.txt says why)
            img = getattr(pin, "_mlp_bwd_images", None)
            if img is None:
                img = pin._mlp_bwd_images = MlpBwdImages(pin, pout, pout.weight.shape[1])
            du, dx = _hip.mlp_bwd(dy2, u, img.operand(), pout.weight.shape[1])
        else:
            du = _hip.linear_swiglu_bwd_bf16(dy2, pout.transposed(), u)
            if deep256_usable(du.shape[0], w1.shape[1], w1.shape[0]):
                dx = deep256(du, pin, True, None)
            elif own_gemm(du.shape[0], w1.shape[1], w1.shape[0]):
                dx = _hip.linear_bf16(du, pin.transposed(), None)
            else:
                dx = du @ w1
        g1 = _pack_weight_grads(du, x2, pin, pin.grad_row_map, None if pin.grad_rows is None else pin.grad_rows.numel())
        g2 = _pack_weight_grads(dy2, s_, pout)
        return (dx.reshape(ctx.xshape), None, None, None, *g1, *g2)


def swiglu_mlp_usable(x: Tensor, width: int) -> bool:
    rows = x.numel() // x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and own_gemm(rows, 2 * width, x.shape[-1], _hip.EPI_SWIGLU)
            and own_gemm(rows, width, x.shape[-1], _hip.EPI_SWIGLU_BWD))


def swiglu_mlp(x: Tensor, pin: PackedWeight, pout: PackedWeight) -> Tensor:
    # grad mode is always off inside Function.forward: whether a backward can follow is decided here
    train = torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in pin.params + pout.params))
    return _SwiGLUMLP.apply(x, pin, pout, train, *pin.params, *pout.params)


# VSDE_ATTN_FUSED_TRAIN=0: keep the separate qk_norm_rope / gate_merge passes in the training step (A/B runs)
ATTN_FUSED_TRAIN = os.environ.get("VSDE_ATTN_FUSED_TRAIN", "1") != "0"
# VSDE_GATE_BWD_GEMM=0: output projection's input gradient and the gate backward as two kernels (A/B runs)
GATE_BWD_GEMM = os.environ.get("VSDE_GATE_BWD_GEMM", "1") != "0"
_NONZERO: dict[int, tuple] = {}   # id(weight) -> (weakref, version, all entries non-zero)


def _all_nonzero(w: Tensor) -> bool:
    """The fused backward divides by the (frozen) RMS weights: checked once per weight version (one host sync)."""
    hit = _NONZERO.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    ok = bool((w.detach() != 0).all())
    _NONZERO[id(w)] = (weakref.ref(w), w._version, ok)
    return ok


class _AttentionCore(torch.autograd.Function):
    """x -> [q | k | v | gate] projection -> QK-RMS-norm, RoPE, value mix -> softmax(q k^T) v -> sigmoid gate, head merge
    (reference primitives/attn.py:80-113) as TWO kernels forward (``vsde_linear_qknorm_bf16``: everything up to the attention
    inputs in the GEMM epilogue; ``vsde_attention_fwd_gated_bf16``: the gate in the attention store; + the output projection when
    its pack is handed in) and three + the projections' GEMMs backward (``vsde_linear_gate_bwd_bf16``: the output projection's
    input gradient with the gate backward in its epilogue -- or ``vsde_gate_bwd_delta`` --, the dq and dk/dv attention kernels
    whose epilogues undo RoPE / RMS-norm / the value mix and write the projection's gradient buffer directly).  The raw projection [B,N,3C+d] is never materialised; the
    backward works from the attention inputs, one inverse RMS per (token, head) and v_raw - v0.

    Value-residual gradient: as in ``_QkNormRopeJoint`` -- consumer blocks accumulate into ``v0link`` inside their dk/dv kernel,
    the producing block (``v0 is None``) folds that buffer into its own dv."""

    @staticmethod
    def forward(ctx, x, pack, opack, cos, sin, wq, wk, v0, lam, heads, eps, scale, v0link, *params):
        """``opack``: the output projection (or None: the merged rows are returned and the caller projects them)."""
        B, N, K = x.shape
        w, b = pack.operands()
        mix = v0 is not None
        v0c = v0.to(torch.bfloat16).contiguous() if mix else None
        lamc = lam.detach().float().reshape(1).contiguous() if mix else None
        x2 = x.reshape(B * N, K)
        q, k, v, glog, rinv, vdiff = _hip.linear_qknorm_bf16(x2, w, b, heads, N, cos, sin, wq, wk, v0c, lamc, eps, save=True)
        shape = (B, N, heads, 64)
        og, lse = _hip.attention_fwd_gated(q.view(shape), k.view(shape), v.view(shape), glog, scale)
        ctx.save_for_backward(x2, q, k, v, glog, og, lse, rinv, vdiff, cos, sin, wq, wk, lamc)
        ctx.meta = (pack, opack, heads, scale, lam.dtype if mix else None, v0link, (B, N, K))
        ctx.set_materialize_grads(False)
        if opack is None:
            return og.view(B, N, heads * 64), v.view(shape)
        wo, bo = opack.operands()
        return _mm_nt(og.view(B * N, heads * 64), wo, bo).view(B, N, wo.shape[0]), v.view(shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dv_out):
        x2, q, k, v, glog, og, lse, rinv, vdiff, cos, sin, wq, wk, lamc = ctx.saved_tensors
        pack, opack, heads, scale, lam_dtype, v0link, (B, N, K) = ctx.meta
        M, C, C3, G = B * N, heads * 64, 3 * heads * 64, glog.shape[1]
        shape = (B, N, heads, 64)
        mix = vdiff is not None
        dy = torch.empty(M, C3 + G, device=x2.device, dtype=torch.bfloat16)
        ograds: tuple = ()
        if opack is None:
            dmerged = torch.zeros(M, C, device=x2.device, dtype=torch.bfloat16) if dout is None else dout.to(torch.bfloat16).reshape(M, C)
            dattn, delta = _hip.gate_bwd_delta(dmerged.contiguous().view(shape), og, glog, dy[:, C3:])
        else:
            wo = opack.weight
            do2 = (torch.zeros(M, wo.shape[0], device=x2.device, dtype=torch.bfloat16) if dout is None
                   else dout.to(torch.bfloat16).reshape(M, wo.shape[0]).contiguous())
            if GATE_BWD_GEMM and wo.shape[0] in (128, 256) and own_gemm(M, C, wo.shape[0]):
                # the projection's input gradient with the gate backward in its epilogue: d(merged) is never written
                dattn, delta = _hip.linear_gate_bwd(do2, opack.transposed(), og, glog, dy[:, C3:], N)
            else:
                dattn, delta = _hip.gate_bwd_delta((do2 @ wo).view(shape), og, glog, dy[:, C3:])
            ograds = tuple(_pack_weight_grads(do2, og.view(M, C), opack))
        acc = v0link.value if (v0link is not None and mix) else None
        extra = v0link.take() if (v0link is not None and not mix) else None
        if dv_out is not None:   # the values were consumed outside the link protocol (a direct use of the returned tensor)
            dv_out = dv_out.to(torch.bfloat16).contiguous()
            extra = dv_out if extra is None else (extra + dv_out)
        dv0, dlam = _hip.attention_bwd_fused(dattn, q.view(shape), k.view(shape), v.view(shape), lse, delta, rinv, cos, sin, wq, wk,
                                             vdiff, lamc, acc, extra, dy, scale)
        if v0link is not None and mix:
            v0link.value, dv0 = dv0, None
        if dlam is not None:
            dlam = dlam.to(lam_dtype).reshape(())
        wb = pack.weight
        dx = None
        if ctx.needs_input_grad[0]:
            if deep256_usable(M, wb.shape[1], wb.shape[0]):
                dx = deep256(dy, pack, True, None).view(B, N, K)
            else:
                dx = (_hip.linear_bf16(dy, pack.transposed(), None) if own_gemm(M, wb.shape[1], wb.shape[0]) else dy @ wb).view(B, N, K)
        return (dx, None, None, None, None, None, None, dv0, dlam, None, None, None, None, *_pack_weight_grads(dy, x2.contiguous(), pack), *ograds)


def attention_core_usable(x: Tensor, pack: "PackedWeight", heads: int, d: int, wq: Tensor, wk: Tensor, cos: Tensor) -> bool:
    """Training-step form of the attention block's core (see ``_AttentionCore``): K in (128, 256), head_dim 64, a 64-wide gate block,
    a sequence the LDS-resident attention kernels take, fp32 non-zero norm weights and fp32 rotary tables."""
    rows = x.numel() // x.shape[-1]
    return (ENABLED and OWN_GEMM and ATTN_FUSED_TRAIN and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16
            and x.ndim == 3 and x.shape[-1] in (128, 256) and d == 64 and pack.weight.shape[0] == 3 * heads * 64 + 64 and rows >= OWN_GEMM_MIN_ROWS
            and wq.dtype == torch.float32 and wk.dtype == torch.float32 and cos.dtype == torch.float32 and cos.shape[-1] == 32
            and _hip.attention_fused_supported(x.shape[1], d) and _all_nonzero(wq) and _all_nonzero(wk))


def attention_core(x: Tensor, pack: "PackedWeight", cos: Tensor, sin: Tensor, wq: Tensor, wk: Tensor, v0: Optional[Tensor],
                   lam: Optional[Tensor], heads: int, eps: float, scale: float, v0link: Optional[GradLink],
                   out_pack: Optional["PackedWeight"] = None):
    """(merged gated attention output [B,N,heads*64] -- or, with ``out_pack``, its output projection --, values [B,N,heads,64])
    for x [B,N,K]; v0 token-major or None.  With ``out_pack`` the projection's input gradient and the gate backward are one
    kernel in the backward (``vsde_linear_gate_bwd_bf16``)."""
    oparams = () if out_pack is None else tuple(out_pack.params)
    return _AttentionCore.apply(x, pack, out_pack, cos.contiguous(), sin.contiguous(), wq.contiguous(), wk.contiguous(), v0,
                                lam if v0 is not None else None, heads, eps, scale, v0link, *pack.params, *oparams)


def row_pack(weights: list[Tensor], biases: Optional[list[Tensor]], pad_to: Optional[int] = None) -> PackedWeight:
    """Pack of several Linears that share their input, stacked along the output rows ([qkv | gate])."""
    rows = sum(w.shape[0] for w in weights)
    wp, bp, r = [], [], 0
    for i, w in enumerate(weights):
        wp.append((w, 0, w.shape[0], r))
        if biases is not None:
            bp.append((biases[i], 0, w.shape[0], r))
        r += w.shape[0]
    return PackedWeight(pad_to or rows, weights[0].shape[1], wp, bp or None, weights[0].device)


def swiglu_packs(w_in: Tensor, b_in: Optional[Tensor], w_out: Tensor, b_out: Optional[Tensor], width: int,
                 interleave: bool = False):
    """Packs of a SwiGLU MLP whose hidden size h is zero-padded to ``width``: input projection [2h, K] -> [2*width, K] (the
    two halves start at rows 0 and ``width``), output projection [K, h] -> [K, width] (extra columns zero).
    ``interleave``: the input projection's halves alternate in blocks of 16 rows ([a_0..15 | b_0..15 | a_16..31 | ...]), the
    layout the fused-SwiGLU GEMM epilogues work on (csrc/vsde_linear.hip)."""
    h = w_out.shape[1]
    if interleave:
        wp, bp, rows = [], [], []
        for half in (0, 1):
            for j0 in range(0, h, 16):
                n, dst = min(16, h - j0), 32 * (j0 // 16) + 16 * half
                wp.append((w_in, half * h + j0, n, dst))
                if b_in is not None:
                    bp.append((b_in, half * h + j0, n, dst))
                rows.extend(range(dst, dst + n))
        pin = PackedWeight(2 * width, w_in.shape[1], wp, bp or None, w_in.device,
                           grad_rows=torch.tensor(rows, device=w_in.device, dtype=torch.long))
    else:
        pin = PackedWeight(2 * width, w_in.shape[1], [(w_in, 0, h, 0), (w_in, h, h, width)],
                           None if b_in is None else [(b_in, 0, h, 0), (b_in, h, h, width)], w_in.device)
    n = w_out.shape[0]
    pout = PackedWeight(n, width, [(w_out, 0, n, 0)], None if b_out is None else [(b_out, 0, n, 0)], w_out.device)
    return pin, pout
