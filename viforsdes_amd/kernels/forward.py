"""``launch_fwd``: host launcher of the fused forward (reference: kernels/forward.py:378-563)."""
from __future__ import annotations

from typing import Optional

from torch import Tensor

from .backend import get_backend
from .weights import SavedActivations, SDEWeights


def launch_fwd(x0: Tensor, context: Tensor, sde_parameters: Tensor, eps: Tensor, weights: SDEWeights,
               time_step: float, save_activations: bool
               ) -> tuple[Tensor, Tensor, Tensor, Optional[SavedActivations]]:
    """Sample B Euler-Maruyama paths through the GRU head.

    ``context`` is ``[B, T, C]`` (fp32 or bf16; the strided ``ctx[:, :-1]`` view is read in
    place).  Returns ``(paths[B,T+1,S], means[B,T,S], cholesky[B,T,S,S], saved)`` in fp32;
    ``saved`` is ``None`` unless ``save_activations``.
    """
    paths, means, chol, chol_raw, acts = get_backend().head_forward(
        x0, context, sde_parameters, eps.reshape(x0.shape[0], -1, x0.shape[1]), weights.tensors(),
        float(time_step), bool(save_activations))
    saved = SavedActivations.from_packed(paths, chol_raw, acts) if save_activations else None
    return paths, means, chol, saved
