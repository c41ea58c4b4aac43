"""Autograd boundary of the fused head (reference: kernels/autograd.py:35-268).

``_SDEFunction.apply`` takes the same 20 positional arguments as the reference and returns
``(diffusion_paths, transition_means, transition_cholesky)``; the backward is once-differentiable
and returns gradients cast to each input's dtype.
"""
from __future__ import annotations

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from .backward import launch_bwd
from .forward import launch_fwd
from .weights import SDEWeights


def dt_ok(dtype: torch.dtype) -> bool:
    return dtype in (torch.float32, torch.bfloat16)


def _as(t: Tensor, dtype: torch.dtype) -> Tensor:
    return t if t.dtype == dtype else t.to(dtype)


class _SDEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, context, sde_parameters, standard_noise, time_step, hidden_dim, context_dim,
                sde_param_dim, state_dim, num_layers, W_ih_l0, W_hh_l0, b_ih_l0, b_hh_l0, W_ih_stack,
                W_hh_stack, b_ih_stack, b_hh_stack, out_weight, out_bias, context_has_extra_step=False):
        # context_has_extra_step (additive, default = the reference's signature): ``context`` is the encoder output
        # [B, T+1, C] itself; the head reads its first T steps in place and the backward writes the context gradient
        # straight into a buffer of that shape and dtype (no cast, no zero-padded copy for the slice).
        ctx.full_context = bool(context_has_extra_step)
        ctx.n_inputs = 21 if context_has_extra_step else 20
        if context_has_extra_step:
            context = context[:, :-1]
        weights = SDEWeights.from_tensors(
            W_ih_l0.detach(), W_hh_l0.detach(), b_ih_l0.detach(), b_hh_l0.detach(), W_ih_stack.detach(),
            W_hh_stack.detach(), b_ih_stack.detach(), b_hh_stack.detach(), out_weight.detach(),
            out_bias.detach(), hidden_dim, context_dim, sde_param_dim, state_dim, num_layers)
        paths, means, chol, saved = launch_fwd(x0.detach(), context.detach(), sde_parameters.detach(),
                                               standard_noise.detach(), weights, time_step, True)
        assert saved is not None
        ctx.save_for_backward(context.detach(), sde_parameters.detach(), standard_noise.detach(),
                              *weights.tensors(), saved.transition_cholesky_raw, saved.packed_activations, paths)
        ctx.meta = (time_step, hidden_dim, context_dim, sde_param_dim, state_dim, num_layers)
        ctx.in_dtypes = (x0.dtype, context.dtype, sde_parameters.dtype) + tuple(
            w.dtype for w in (W_ih_l0, W_hh_l0, b_ih_l0, b_hh_l0, W_ih_stack, W_hh_stack, b_ih_stack,
                              b_hh_stack, out_weight, out_bias))
        out_dtype = x0.dtype
        return _as(paths, out_dtype), _as(means, out_dtype), _as(chol, out_dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_paths, g_means, g_chol):
        from .weights import SavedActivations
        saved_t = ctx.saved_tensors
        context, theta, noise = saved_t[:3]
        wts = saved_t[3:13]
        chol_raw, acts, paths = saved_t[13:16]
        time_step, H, C, P, S, L = ctx.meta
        weights = SDEWeights.from_tensors(*wts, H, C, P, S, L)
        saved = SavedActivations.from_packed(paths, chol_raw, acts)
        gfull = None
        if ctx.full_context and dt_ok(ctx.in_dtypes[1]):
            B, T, C = context.shape
            gfull = torch.empty(B, T + 1, C, device=context.device, dtype=ctx.in_dtypes[1])
            gfull[:, T].zero_()  # the last token never reaches the head
        grads = launch_bwd(g_paths.float(), g_means.float(), g_chol.float(), context, theta, noise, saved,
                           weights, time_step, context_grad_out=gfull)
        dt = ctx.in_dtypes
        gx0, gth = _as(grads[0], dt[0]), _as(grads[2], dt[2])
        if gfull is not None:
            gctx = gfull
        else:
            gctx = _as(grads[1], dt[1])
            if ctx.full_context:
                gctx = torch.nn.functional.pad(gctx, (0, 0, 0, 1))
        gw = [_as(g, d) for g, d in zip(grads[3:], dt[3:])]
        if L == 1:  # empty [0, 3H, H] stacks
            for i in (4, 5, 6, 7):
                gw[i] = torch.zeros_like(wts[i])
        return (gx0, gctx, gth, None, None, None, None, None, None, None, *gw) + ((None,) if ctx.n_inputs == 21 else ())


def sample_diffusion_paths(x0: Tensor, context: Tensor, sde_parameters: Tensor, standard_noise: Tensor,
                           weights: SDEWeights, time_step: float) -> tuple[Tensor, Tensor, Tensor]:
    """No-grad sampling launch (reference: kernels/autograd.py:244-268)."""
    paths, means, chol, _ = launch_fwd(x0, context, sde_parameters, standard_noise, weights, time_step, False)
    d = x0.dtype
    return _as(paths, d), _as(means, d), _as(chol, d)
