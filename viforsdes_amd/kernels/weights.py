"""Weight and activation containers of the fused head.

Mirrors ``SDEWeights`` / ``SavedActivations`` of the reference (kernels/weights.py:11-196) at
the interface level: same constructors, same field names.  Two deliberate differences:

* the reference re-transposes every matrix to ``[in, 3H]`` on each forward AND backward
  (kernels/autograd.py:62-78,169-185); the HIP kernels consume torch.nn.GRU's native
  ``[3H, in]`` tensors directly (they re-pack on the device in a few microseconds), so the
  fields here hold the native layout and no copy is made;
* the twelve saved tensors are views into two packed buffers (``packed_cholesky_raw``,
  ``packed_activations[B, T, L, 5, H]``) so that the backward streams one contiguous record
  per path-step.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
from torch import Tensor, nn

from .constants import MAX_LAYERS


class SavedActivations(NamedTuple):
    diffusion_paths: Tensor           # [B, T+1, S]
    transition_cholesky_raw: Tensor   # [B, T, n_tril]
    h_l0: Tensor                      # [B, T, H]   (views into packed_activations)
    r_l0: Tensor
    z_l0: Tensor
    n_l0: Tensor
    n_hh_l0: Tensor
    h_stack: Tensor                   # [B, L-1, T, H]
    r_stack: Tensor
    z_stack: Tensor
    n_stack: Tensor
    n_hh_stack: Tensor
    packed_activations: Tensor        # [B, T, L, 5, H]  slots (h, r, z, n, n_hh)

    @classmethod
    def from_packed(cls, paths: Tensor, chol_raw: Tensor, acts: Tensor) -> "SavedActivations":
        l0 = [acts[:, :, 0, k, :] for k in range(5)]
        if acts.shape[2] > 1:
            st = [acts[:, :, 1:, k, :].permute(0, 2, 1, 3) for k in range(5)]
        else:
            st = [acts.new_empty(0) for _ in range(5)]
        return cls(paths, chol_raw, *l0, *st, acts)


class SDEWeights:
    """The ten head tensors plus dimensions (nn.GRU-native layout, see module docstring)."""

    __slots__ = ("W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0", "W_ih_stack", "W_hh_stack",
                 "b_ih_stack", "b_hh_stack", "out_weight", "out_bias", "hidden_dim",
                 "context_dim", "sde_param_dim", "state_dim", "num_layers")

    def __init__(self, W_ih_l0: Tensor, W_hh_l0: Tensor, b_ih_l0: Tensor, b_hh_l0: Tensor,
                 W_ih_stack: Tensor, W_hh_stack: Tensor, b_ih_stack: Tensor, b_hh_stack: Tensor,
                 out_weight: Tensor, out_bias: Tensor, hidden_dim: int, context_dim: int,
                 sde_param_dim: int, state_dim: int, num_layers: int) -> None:
        if num_layers < 1 or num_layers > MAX_LAYERS:
            raise ValueError(f"num_layers must be in [1, {MAX_LAYERS}], got {num_layers}")
        expect = (3 * hidden_dim, state_dim + context_dim + sde_param_dim)
        if tuple(W_ih_l0.shape) != expect:
            raise ValueError(f"W_ih_l0 must have shape {expect}, got {tuple(W_ih_l0.shape)}")
        self.W_ih_l0, self.W_hh_l0, self.b_ih_l0, self.b_hh_l0 = W_ih_l0, W_hh_l0, b_ih_l0, b_hh_l0
        self.W_ih_stack, self.W_hh_stack = W_ih_stack, W_hh_stack
        self.b_ih_stack, self.b_hh_stack = b_ih_stack, b_hh_stack
        self.out_weight, self.out_bias = out_weight, out_bias
        self.hidden_dim, self.context_dim = hidden_dim, context_dim
        self.sde_param_dim, self.state_dim, self.num_layers = sde_param_dim, state_dim, num_layers

    def tensors(self) -> tuple[Tensor, ...]:
        return (self.W_ih_l0, self.W_hh_l0, self.b_ih_l0, self.b_hh_l0, self.W_ih_stack,
                self.W_hh_stack, self.b_ih_stack, self.b_hh_stack, self.out_weight, self.out_bias)

    @classmethod
    def from_tensors(cls, W_ih_l0: Tensor, W_hh_l0: Tensor, b_ih_l0: Tensor, b_hh_l0: Tensor,
                     W_ih_stack: Tensor, W_hh_stack: Tensor, b_ih_stack: Tensor, b_hh_stack: Tensor,
                     out_weight: Tensor, out_bias: Tensor, hidden_dim: int, context_dim: int,
                     sde_param_dim: int, state_dim: int, num_layers: int) -> "SDEWeights":
        return cls(W_ih_l0, W_hh_l0, b_ih_l0, b_hh_l0, W_ih_stack, W_hh_stack, b_ih_stack,
                   b_hh_stack, out_weight, out_bias, hidden_dim, context_dim, sde_param_dim,
                   state_dim, num_layers)

    @classmethod
    def from_modules(cls, gru: nn.GRU, out_proj: nn.Linear, context_dim: int, sde_param_dim: int,
                     state_dim: int) -> "SDEWeights":
        L, H = gru.num_layers, gru.hidden_size
        if L > MAX_LAYERS:
            raise ValueError(f"num_layers must be <= {MAX_LAYERS}, got {L}")
        p0 = gru.weight_ih_l0
        det = lambda name: getattr(gru, name).detach()
        if L > 1:
            stacks = [torch.stack([det(f"{kind}_l{k}") for k in range(1, L)])
                      for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        else:
            stacks = [p0.new_empty(0, 3 * H, H), p0.new_empty(0, 3 * H, H),
                      p0.new_empty(0, 3 * H), p0.new_empty(0, 3 * H)]
        assert out_proj.bias is not None
        return cls(det("weight_ih_l0"), det("weight_hh_l0"), det("bias_ih_l0"), det("bias_hh_l0"),
                   *stacks, out_proj.weight.detach(), out_proj.bias.detach(), H, context_dim,
                   sde_param_dim, state_dim, L)
