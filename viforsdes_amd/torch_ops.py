"""The fused head behind the torch dispatcher: ``torch.ops.vsde.sde_fwd`` / ``torch.ops.vsde.sde_bwd``.

SURVEY.md section 8(b) words the drop-in boundary as operators registered with ``TORCH_LIBRARY``; the product boundary of
this package is the torch-free C ABI (``include/vsde_hip.h``), and this module is the thin registration on top of it for
callers that want dispatcher-visible operators (``torch.library`` is the Python front end of ``TORCH_LIBRARY``).  The
operators take and return plain tensors in the order of the reference's launchers (``kernels/forward.py:378-563``,
``kernels/backward.py:627-784``); the ten weight tensors travel as one list in ``SDEWeights`` order (``kernels/weights.py:79``).

Importing the module registers the operators; nothing else in the package depends on it."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _hip


@torch.library.custom_op("vsde::sde_fwd", mutates_args=())
def sde_fwd(x0: Tensor, context: Tensor, sde_parameters: Tensor, eps: Tensor, weights: list[Tensor], time_step: float,
            save_activations: bool) -> tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(paths [B,T+1,S], means [B,T,S], cholesky [B,T,S,S], cholesky_raw [B,T,ntril], activations [B,T,L,5,H]); the last
    two are empty tensors unless ``save_activations``."""
    paths, means, chol, raw, acts = _hip.head_forward(x0, context, sde_parameters, eps, list(weights), float(time_step),
                                                      bool(save_activations))
    # custom-op outputs must not alias each other: two separate empty tensors
    return (paths, means, chol, raw if raw is not None else x0.new_empty(0), acts if acts is not None else x0.new_empty(0))


@sde_fwd.register_fake
def _(x0, context, sde_parameters, eps, weights, time_step, save_activations):
    B, S = x0.shape
    T = context.shape[1]
    H, L = weights[1].shape[1], 1 + weights[4].shape[0]
    f32 = dict(device=x0.device, dtype=torch.float32)
    ntril = S * (S + 1) // 2
    return (torch.empty(B, T + 1, S, **f32), torch.empty(B, T, S, **f32), torch.empty(B, T, S, S, **f32),
            torch.empty((B, T, ntril) if save_activations else (0,), **f32),
            torch.empty((B, T, L, 5, H) if save_activations else (0,), **f32))


@torch.library.custom_op("vsde::sde_bwd", mutates_args=())
def sde_bwd(grad_paths: Tensor, grad_means: Tensor, grad_cholesky: Tensor, context: Tensor, sde_parameters: Tensor, eps: Tensor,
            paths: Tensor, cholesky_raw: Tensor, activations: Tensor, weights: list[Tensor], time_step: float) -> list[Tensor]:
    """The 13 fp32 gradients in the reference's order: x0, context, sde_parameters, then the ten weights."""
    return list(_hip.head_backward(grad_paths, grad_means, grad_cholesky, context, sde_parameters, eps, paths, cholesky_raw,
                                   activations, list(weights), float(time_step)))


@sde_bwd.register_fake
def _(grad_paths, grad_means, grad_cholesky, context, sde_parameters, eps, paths, cholesky_raw, activations, weights, time_step):
    f32 = dict(device=paths.device, dtype=torch.float32)
    B, S = paths.shape[0], paths.shape[2]
    return ([torch.empty(B, S, **f32), torch.empty(context.shape, **f32), torch.empty(sde_parameters.shape, **f32)]
            + [torch.empty(w.shape, **f32) for w in weights])
