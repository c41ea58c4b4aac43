"""Monte-Carlo ELBO (reference: inference/evidence_lower_bound.py:19-83).

    ELBO = mean_b( obs_lp + sde_lp - gen_lp + log_jacobian + log p(theta) - log q(theta) )

The three B*T-sized terms (SDE transition log-density, generative/variational path log-density,
softplus log-Jacobian) are one fused HIP kernel with an analytic backward
(csrc/vsde_elbo.hip); drift and diffusion are the user's Python callables evaluated on the
flattened ``[(B*T), S]`` states exactly like the reference does (lines 37-40) -- except for the SDEs
built into the HIP library (``sde.builtin_kind``), whose coefficients and vector-Jacobian products are
one kernel each (csrc/vsde_sde.hip)."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from ..core.observations import ObservationLikelihood, Observations, grid_index
from ..core.priors import Prior
from ..core.sde import SDE, builtin_sde_kind
from ..kernels.backend import get_backend
from ..models.sde_parameter_posterior import SDEParameterPosterior
from .types import DiffusionPathSample, EvidenceLowerBoundComponents, EvidenceLowerBoundResult


class _PathTerms(torch.autograd.Function):
    """(z, x, means, chol, drift, diffusion) -> (sde_lp[B], gen_lp[B], log_jac[B])."""

    @staticmethod
    def forward(ctx, z, x, means, chol, drift, diffusion, positive_dims, time_step):
        args = tuple(t.detach().float().contiguous() for t in (z, x, means, chol, drift, diffusion))
        ctx.save_for_backward(*args)
        ctx.misc = (tuple(positive_dims), float(time_step), tuple(t.dtype for t in (z, x, means, chol, drift, diffusion)))
        return get_backend().elbo_path_terms(*args, list(positive_dims), float(time_step))

    @staticmethod
    @once_differentiable
    def backward(ctx, g_sde, g_gen, g_jac):
        positive_dims, time_step, dtypes = ctx.misc
        grads = get_backend().elbo_path_terms_bwd(*ctx.saved_tensors, list(positive_dims), time_step,
                                                  g_sde.float().contiguous(), g_gen.float().contiguous(),
                                                  g_jac.float().contiguous())
        return tuple(g.to(d) for g, d in zip(grads, dtypes)) + (None, None)


HIP_COEFFICIENTS = True  # set False to evaluate built-in SDEs through their Python callables too (A/B tests)


class _BuiltinCoefficients(torch.autograd.Function):
    """(x [B,T+1,S], theta [B,P]) -> (drift [B,T,S], diffusion [B,T,S,S]) of a built-in SDE: one kernel forward, one backward
    (csrc/vsde_sde.hip) instead of the ~100 tiny kernels the Python callables and their autograd graph expand to."""

    @staticmethod
    def forward(ctx, x, theta, kind):
        from .. import _hip
        xc, tc = x.detach().float().contiguous(), theta.detach().float().contiguous()
        ctx.save_for_backward(xc, tc)
        ctx.meta = (kind, x.dtype, theta.dtype)
        return _hip.sde_coefficients_fwd(kind, xc, tc)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_drift, g_diffusion):
        from .. import _hip
        kind, xdtype, tdtype = ctx.meta
        g_x, g_theta = _hip.sde_coefficients_bwd(kind, *ctx.saved_tensors, g_drift.float().contiguous(),
                                                 g_diffusion.float().contiguous())
        return g_x.to(xdtype), g_theta.to(tdtype), None


def sde_coefficients(sde: SDE, x: Tensor, sde_parameters: Tensor) -> tuple[Tensor, Tensor]:
    """Drift ``[B,T,S]`` and diffusion ``[B,T,S,S]`` on the first T grid points of ``x [B,T+1,S]`` (reference lines 37-40)."""
    B, n_steps, S = x.shape[0], x.shape[1] - 1, x.shape[2]
    kind = builtin_sde_kind(sde)
    if kind is not None and HIP_COEFFICIENTS and x.is_cuda and x.dtype == torch.float32:
        from .. import _hip
        if kind in _hip.SDE_KINDS:
            return _BuiltinCoefficients.apply(x, sde_parameters, kind)
    x_flat = x[:, :-1].reshape(B * n_steps, S)
    theta_flat = sde_parameters.unsqueeze(1).expand(B, n_steps, -1).reshape(B * n_steps, -1)
    return (sde.drift(x_flat, theta_flat).reshape(B, n_steps, S),
            sde.diffusion(x_flat, theta_flat).reshape(B, n_steps, S, S))


HIP_TAIL = True  # set False to keep the [B]-sized tail of the ELBO in torch ops (A/B tests)


class _ElboTail(torch.autograd.Function):
    """Observation / prior / posterior log-densities and the batch means of the ELBO and its components as one kernel
    (csrc/vsde_elbo.hip: elbo_tail_*_kernel) -> ``[elbo, obs, sde, gen, prior, post]``."""

    @staticmethod
    def forward(ctx, x_obs, theta, post_mean, post_log_std, sde_lp, gen_lp, jac, cfg):
        from .. import _hip
        obs_values, obs_matrix, variance, prior_type, prior_mean, prior_std, theta_pos = cfg
        tens = tuple(t.detach().float().contiguous() for t in (x_obs, theta, post_mean, post_log_std))
        ctx.save_for_backward(*tens)
        ctx.cfg = cfg
        ctx.dtypes = tuple(t.dtype for t in (x_obs, theta, post_mean, post_log_std, sde_lp, gen_lp, jac))
        return _hip.elbo_tail_fwd(tens[0], obs_values, obs_matrix, variance, tens[1], prior_type, prior_mean, prior_std, tens[2],
                                  tens[3], theta_pos, sde_lp.detach(), gen_lp.detach(), jac.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        from .. import _hip
        obs_values, obs_matrix, variance, prior_type, prior_mean, prior_std, theta_pos = ctx.cfg
        x_obs, theta, post_mean, post_log_std = ctx.saved_tensors
        grads = _hip.elbo_tail_bwd(x_obs, obs_values, obs_matrix, variance, theta, prior_type, prior_mean, prior_std, post_mean,
                                   post_log_std, theta_pos, g_out.float().contiguous())
        return tuple(g.to(d) for g, d in zip(grads, ctx.dtypes)) + (None,)


def _fused_tail_config(observations: Observations, observation_likelihood, prior, sde_parameter_posterior, x: Tensor,
                       sde_parameters: Tensor):
    """Arguments of the fused tail kernel when every piece is one of the package's own closed forms, else None."""
    from ..core.observations import GaussianObservationLikelihood
    from ..core.priors import PriorType
    if not (HIP_TAIL and x.is_cuda and x.dtype == torch.float32 and sde_parameters.dtype == torch.float32):
        return None
    if type(observation_likelihood) is not GaussianObservationLikelihood or type(prior) is not Prior \
            or type(sde_parameter_posterior) is not SDEParameterPosterior:
        return None
    H = observation_likelihood.obs_matrix
    S, O, P = x.shape[-1], observations.values.shape[-1], sde_parameters.shape[-1]
    if max(S, O, P) > 16 or (H is None and O != S) or (H is not None and tuple(H.shape) != (O, S)) or prior.dim != P:
        return None
    theta_pos = getattr(sde_parameter_posterior, "_positive_dims", None)
    if theta_pos is None:
        return None
    return (observations.values, None if H is None else H.to(x), float(observation_likelihood.variance),
            1 if prior.type == PriorType.LOG_NORMAL else 0, float(prior.mean), float(prior.std), tuple(theta_pos))


def path_log_terms(sample: DiffusionPathSample, drift: Tensor, diffusion: Tensor, time_step: float
                   ) -> tuple[Tensor, Tensor, Tensor]:
    """Per-sample ``(sde_log_prob, generative_log_prob, log_jacobian)``, each ``[B]``."""
    return _PathTerms.apply(sample.z, sample.x, sample.transition_means, sample.transition_cholesky, drift,
                            diffusion, sample.state_space.positive_dims, time_step)


def compute_evidence_lower_bound(sde: SDE, observations: Observations, observation_likelihood: ObservationLikelihood,
                                 prior: Prior, sde_parameter_posterior: SDEParameterPosterior, sde_parameters: Tensor,
                                 sample: DiffusionPathSample, time_step: float) -> EvidenceLowerBoundResult:
    z = sample.z
    B, n_steps, S = z.shape[0], z.shape[1] - 1, z.shape[2]
    x = sample.x
    drift, diffusion = sde_coefficients(sde, x, sde_parameters)

    sde_lp, gen_lp, jac = _PathTerms.apply(z, x, sample.transition_means, sample.transition_cholesky, drift,
                                           diffusion, sample.state_space.positive_dims, time_step)

    obs_idx = grid_index(observations.times, time_step, n_steps)
    cfg = _fused_tail_config(observations, observation_likelihood, prior, sde_parameter_posterior, x, sde_parameters)
    if cfg is not None:
        out = _ElboTail.apply(x[:, obs_idx], sde_parameters, sde_parameter_posterior.mean, sde_parameter_posterior.log_std,
                              sde_lp, gen_lp, jac, cfg)
        return EvidenceLowerBoundResult(
            evidence_lower_bound=out[0],
            components=EvidenceLowerBoundComponents(
                observation_log_prob=out[1], sde_log_prob=out[2], generative_log_prob=out[3], prior_log_prob=out[4],
                posterior_log_prob=out[5]))
    obs_lp = observation_likelihood.log_prob(observations.values.unsqueeze(0).expand(B, -1, -1), x[:, obs_idx]).sum(dim=-1)
    prior_lp = prior.log_prob(sde_parameters)
    if prior_lp.ndim > 1:
        prior_lp = prior_lp.sum(dim=-1)
    post_lp = sde_parameter_posterior.log_prob(sde_parameters)

    elbo = obs_lp + sde_lp - gen_lp + jac + prior_lp - post_lp
    return EvidenceLowerBoundResult(
        evidence_lower_bound=elbo.mean(),
        components=EvidenceLowerBoundComponents(
            observation_log_prob=obs_lp.mean(), sde_log_prob=sde_lp.mean(), generative_log_prob=gen_lp.mean(),
            prior_log_prob=prior_lp.mean(), posterior_log_prob=post_lp.mean()))
