"""GRU transition head (reference: models/head.py:20-209).

Parameters live in a ``torch.nn.GRU`` + ``nn.Linear`` so the checkpoint keys match
(``head.gru.weight_ih_l0`` ..., ``head.out_proj.*``); the computation never goes through
``nn.GRU``: ``sample_diffusion_paths`` hands the raw parameter tensors to the fused HIP operator.
``forward`` is the eager single-step definition of the same arithmetic (kept, like in the
reference, as the semantic specification of one kernel step)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from ..config import HeadConfig
from ..inference.constants import DIAG_MIN
from ..kernels.autograd import _SDEFunction, sample_diffusion_paths as _sample_no_grad
from ..kernels.constants import MAX_LAYERS
from ..kernels.weights import SDEWeights
from ..primitives.bounds import lower_bound


class DiffusionTransitionHead(nn.Module):
    _tril_rows: Tensor
    _tril_cols: Tensor
    _diag_mask: Tensor

    def __init__(self, state_dim: int, context_dim: int, sde_param_dim: int, config: HeadConfig) -> None:
        super().__init__()
        if not 1 <= config.num_layers <= MAX_LAYERS:
            raise ValueError(f"num_layers must be in [1, {MAX_LAYERS}], got {config.num_layers}")
        self.state_dim, self.context_dim, self.sde_param_dim = state_dim, context_dim, sde_param_dim
        self.hidden_dim, self.num_layers = config.hidden_dim, config.num_layers
        self.n_tril = state_dim * (state_dim + 1) // 2
        rows, cols = torch.tril_indices(state_dim, state_dim)
        self.register_buffer("_tril_rows", rows)
        self.register_buffer("_tril_cols", cols)
        self.register_buffer("_diag_mask", rows == cols)
        self.gru = nn.GRU(input_size=state_dim + context_dim + sde_param_dim, hidden_size=config.hidden_dim,
                          num_layers=config.num_layers, batch_first=True)
        self.out_proj = nn.Linear(config.hidden_dim, state_dim + self.n_tril)
        with torch.no_grad():  # mu = 0, L = I at initialisation
            self.out_proj.weight.zero_()
            self.out_proj.bias.copy_(torch.cat([torch.zeros(state_dim), self._diag_mask.float()]))

    # ---- eager single step (specification; not used by training) ---------------------------
    def init_hidden(self, batch: int, device: torch.device, dtype: torch.dtype = torch.float32) -> Tensor:
        return torch.zeros(self.num_layers, batch, self.hidden_dim, device=device, dtype=dtype)

    def forward(self, x_t: Tensor, context_t: Tensor, sde_parameters: Tensor, hidden: Tensor | None = None
                ) -> tuple[Tensor, Tensor, Tensor]:
        step_in = torch.cat([x_t, context_t, sde_parameters], dim=-1).unsqueeze(1)
        out, hidden = self.gru(step_in, hidden)
        emitted = self.out_proj(out.squeeze(1))
        mu, tril = emitted[..., :self.state_dim], emitted[..., self.state_dim:]
        entries = torch.where(self._diag_mask, lower_bound(tril, DIAG_MIN), tril)
        L = tril.new_zeros(tril.shape[0], self.state_dim, self.state_dim)
        L[:, self._tril_rows, self._tril_cols] = entries
        return mu, L, hidden

    # ---- fused operator ---------------------------------------------------------------------
    def _extract_gru_weights(self) -> tuple[Tensor, ...]:
        g, H = self.gru, self.hidden_dim
        first = (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        if self.num_layers == 1:
            w = g.weight_ih_l0
            return first + (w.new_empty(0, 3 * H, H), w.new_empty(0, 3 * H, H), w.new_empty(0, 3 * H),
                            w.new_empty(0, 3 * H))
        upper = range(1, self.num_layers)
        if self.num_layers == 2:   # one upper layer: the stacked tensor is a view of the parameter (no copy, no copy-backward)
            return first + tuple(getattr(g, f"{kind}_l1").unsqueeze(0) for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
        return first + tuple(torch.stack([getattr(g, f"{kind}_l{k}") for k in upper])
                             for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))

    accepts_full_context = True  # sample_diffusion_paths(..., context_has_extra_step=True) takes the encoder output unsliced

    def sample_diffusion_paths(self, x0: Tensor, context: Tensor, sde_parameters: Tensor, standard_noise: Tensor,
                               time_step: float, context_has_extra_step: bool = False) -> tuple[Tensor, Tensor, Tensor]:
        """``(paths[B,T+1,S], transition_means[B,T,S], transition_cholesky[B,T,S,S])``.

        ``context_has_extra_step`` (additive): ``context`` is ``[B, T+1, C]`` and only its first T steps are used -- what
        the sampler has anyway (``diffusion_path_sampler.py:66`` slices ``[:, :-1]``); saves the slice's backward copy."""
        if self.training and context_has_extra_step:
            return _SDEFunction.apply(x0, context, sde_parameters, standard_noise, time_step, self.hidden_dim,
                                      self.context_dim, self.sde_param_dim, self.state_dim, self.num_layers,
                                      *self._extract_gru_weights(), self.out_proj.weight, self.out_proj.bias, True)
        if context_has_extra_step:
            context = context[:, :-1]
        if self.training:
            return _SDEFunction.apply(x0, context, sde_parameters, standard_noise, time_step, self.hidden_dim,
                                      self.context_dim, self.sde_param_dim, self.state_dim, self.num_layers,
                                      *self._extract_gru_weights(), self.out_proj.weight, self.out_proj.bias)
        weights = SDEWeights.from_modules(self.gru, self.out_proj, self.context_dim, self.sde_param_dim,
                                          self.state_dim)
        return _sample_no_grad(x0, context, sde_parameters, standard_noise, weights, time_step)
