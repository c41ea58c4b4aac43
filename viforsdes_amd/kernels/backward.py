"""``launch_bwd``: host launcher of the fused backward (reference: kernels/backward.py:627-784)."""
from __future__ import annotations

from torch import Tensor

from .backend import get_backend
from .weights import SavedActivations, SDEWeights


def launch_bwd(grad_diffusion_paths: Tensor, grad_transition_means: Tensor,
               grad_transition_cholesky: Tensor, context: Tensor, sde_parameters: Tensor, eps: Tensor,
               saved: SavedActivations, weights: SDEWeights, time_step: float,
               context_grad_out: "Tensor | None" = None) -> tuple[Tensor, ...]:
    """Returns the 13 fp32 gradients ``(x0, context, sde_parameters, W_ih_l0, W_hh_l0, b_ih_l0,
    b_hh_l0, W_ih_stack, W_hh_stack, b_ih_stack, b_hh_stack, out_weight, out_bias)`` with the weight
    gradients in nn.GRU-native layout, exactly like the reference's return value."""
    B = context.shape[0]
    S = saved.diffusion_paths.shape[2]
    if context_grad_out is not None:  # additive: gradient written in place into a [B, T+1, C] buffer (see _hip.head_backward)
        return get_backend().head_backward(
            grad_diffusion_paths, grad_transition_means, grad_transition_cholesky, context, sde_parameters,
            eps.reshape(B, -1, S), saved.diffusion_paths, saved.transition_cholesky_raw,
            saved.packed_activations, weights.tensors(), float(time_step), context_grad_out=context_grad_out)
    return get_backend().head_backward(
        grad_diffusion_paths, grad_transition_means, grad_transition_cholesky, context, sde_parameters,
        eps.reshape(B, -1, S), saved.diffusion_paths, saved.transition_cholesky_raw,
        saved.packed_activations, weights.tensors(), float(time_step))
