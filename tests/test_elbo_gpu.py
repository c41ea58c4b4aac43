"""GPU parity of the fused ELBO path-term kernels and of the whole ELBO/autograd chain.

Tolerances: per-sample terms 5e-6 relative (fp32, v_log/v_exp based), gradients 5e-5 relative-to-max."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, W_NAMES, G_NAMES, load_head_case, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", ["ou", "lv"])
def test_path_terms_forward_backward_vs_oracle_and_golden(name):
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    d = dict(np.load(f"{GOLDEN}/elbo_{name}.npz"))
    B, T, S, P = (int(v) for v in d["dims"])
    sp, dt = [int(v) for v in d["state_positive_dims"]], float(d["dt"])
    args = [_t(d[k]) for k in ("z", "x", "means", "chol", "drift", "diffusion")]
    s, g, j = _hip.elbo_path_terms(*args, sp, dt)
    assert rel_err(s.cpu().numpy(), d["sde_lp"]) < 5e-6 and rel_err(g.cpu().numpy(), d["gen_lp"]) < 5e-6
    if d["jac"].any():
        assert rel_err(j.cpu().numpy(), d["jac"]) < 5e-6
    rng = np.random.default_rng(1)
    gs, gg, gj = (rng.normal(size=B).astype(np.float32) for _ in range(3))
    got = _hip.elbo_path_terms_bwd(*args, sp, dt, _t(gs), _t(gg), _t(gj))
    ref = vo.elbo_path_terms_bwd(d["z"], d["x"], d["means"], d["chol"], d["drift"], d["diffusion"], sp, dt, gs, gg, gj,
                                 np.float64)
    for a, b_, nm in zip(got, ref, ("z", "x", "means", "chol", "drift", "diffusion")):
        assert rel_err(a.cpu().numpy(), b_) < 5e-5, nm


@pytest.mark.parametrize("name", ["ou", "lv"])
def test_compute_evidence_lower_bound_and_autograd_vs_golden(name):
    from viforsdes_amd import GaussianObservationLikelihood, Observations, Prior, PriorType
    from viforsdes_amd.examples.sdes import LotkaVolterra, OrnsteinUhlenbeck
    from viforsdes_amd.inference.evidence_lower_bound import compute_evidence_lower_bound
    from viforsdes_amd.inference.state_space import StateSpace
    from viforsdes_amd.inference.types import DiffusionPathSample
    from viforsdes_amd.models.sde_parameter_posterior import SDEParameterPosterior
    d = dict(np.load(f"{GOLDEN}/elbo_{name}.npz"))
    B, T, S, P = (int(v) for v in d["dims"])
    sde = LotkaVolterra() if name == "lv" else OrnsteinUhlenbeck()
    post = SDEParameterPosterior(P, [int(v) for v in d["theta_positive_dims"]]).to(DEV)
    with torch.no_grad():
        post.mean.copy_(_t(d["q_mean"])); post.log_std.copy_(_t(d["q_log_std"]))
    prior = Prior(type=PriorType.LOG_NORMAL if int(d["prior_type"]) else PriorType.NORMAL, mean=float(d["prior_mean"]),
                  std=float(d["prior_std"]), dim=P)
    obs = Observations(times=_t(d["obs_times"]), values=_t(d["obs_values"]))
    z, means, chol, theta = (_t(d[k]).requires_grad_(True) for k in ("z", "means", "chol", "theta"))
    sample = DiffusionPathSample(z=z, transition_means=means, transition_cholesky=chol,
                                 state_space=StateSpace(S, [int(v) for v in d["state_positive_dims"]]))
    res = compute_evidence_lower_bound(sde, obs, GaussianObservationLikelihood(variance=float(d["variance"])), prior, post,
                                       theta, sample, float(d["dt"]))
    assert abs(float(res.evidence_lower_bound) - float(d["elbo"])) < 5e-6 * abs(float(d["elbo"]))
    c = res.components
    for got, key in ((c.observation_log_prob, "comp_obs"), (c.sde_log_prob, "comp_sde"), (c.generative_log_prob, "comp_gen"),
                     (c.prior_log_prob, "comp_prior"), (c.posterior_log_prob, "comp_post")):
        assert abs(float(got) - float(d[key])) < 5e-6 * max(1.0, abs(float(d[key]))), key
    grads = torch.autograd.grad(res.evidence_lower_bound, [z, means, chol, theta, post.mean, post.log_std])
    for g, key in zip(grads, ("grad_z", "grad_means", "grad_chol", "grad_theta", "grad_q_mean", "grad_q_log_std")):
        assert rel_err(g.cpu().numpy(), d[key]) < 5e-5, key


def test_path_terms_full_size_properties():
    """LV size: additivity over time (sum of two half-horizon calls = full call) and zero-noise identity."""
    from viforsdes_amd import _hip
    g = torch.Generator().manual_seed(5)
    B, T, S = 512, 400, 2
    rn = lambda *s: torch.randn(*s, generator=g).to(DEV)
    z = rn(B, T + 1, S).cumsum(1) * 0.1
    x = torch.nn.functional.softplus(z)
    means, drift = rn(B, T, S), rn(B, T, S)
    mk = lambda: torch.tril(rn(B, T, S, S) * 0.2, -1) + torch.diag_embed(torch.rand(B, T, S, generator=g).to(DEV) + 0.5)
    chol, diff = mk(), mk()
    full = _hip.elbo_path_terms(z, x, means, chol, drift, diff, [0, 1], 0.1)
    h = T // 2
    a = _hip.elbo_path_terms(z[:, :h + 1].contiguous(), x[:, :h + 1].contiguous(), means[:, :h].contiguous(),
                             chol[:, :h].contiguous(), drift[:, :h].contiguous(), diff[:, :h].contiguous(), [0, 1], 0.1)
    b_ = _hip.elbo_path_terms(z[:, h:].contiguous(), x[:, h:].contiguous(), means[:, h:].contiguous(),
                              chol[:, h:].contiguous(), drift[:, h:].contiguous(), diff[:, h:].contiguous(), [0, 1], 0.1)
    for f, p, q in zip(full, a, b_):
        assert torch.allclose(f, p + q, rtol=2e-5, atol=1e-3)
    # gen term identity: if z_{t+1} = z_t + mu dt + L eps sqrt(dt) then gen_lp = -1/2|eps|^2 - sum log(L_ii sqrt dt) - TS/2 log 2pi
    eps = rn(B, T, S)
    zz = [z[:, 0]]
    for t in range(T):
        zz.append(zz[-1] + means[:, t] * 0.1 + torch.einsum("bij,bj->bi", chol[:, t], eps[:, t]) * 0.1 ** 0.5)
    z2 = torch.stack(zz, 1)
    _, gen, _ = _hip.elbo_path_terms(z2, torch.nn.functional.softplus(z2), means, chol, drift, diff, [0, 1], 0.1)
    want = (-0.5 * (eps ** 2).sum((1, 2)) - torch.log(torch.diagonal(chol, dim1=-2, dim2=-1) * 0.1 ** 0.5).sum((1, 2))
            - T * S / 2 * np.log(2 * np.pi))
    assert torch.allclose(gen, want, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("prior_type,with_matrix,theta_pos", [("normal", False, []), ("log_normal", False, [0, 2]),
                                                              ("log_normal", True, [0, 1, 2]), ("normal", True, [1])])
def test_fused_tail_matches_the_torch_composition(prior_type, with_matrix, theta_pos):
    """Observation / prior / posterior terms + batch means: the single-kernel tail against the same ELBO assembled from the
    package's Python closed forms (which the golden ELBO cases pin to the reference), values and every gradient.  Covers an
    observation matrix (O != S) and both prior families, which the golden cases do not."""
    from viforsdes_amd import GaussianObservationLikelihood, Observations, Prior, PriorType
    from viforsdes_amd.examples.sdes import LotkaVolterra
    from viforsdes_amd.inference import evidence_lower_bound as em
    from viforsdes_amd.inference.state_space import StateSpace
    from viforsdes_amd.inference.types import DiffusionPathSample
    from viforsdes_amd.models.sde_parameter_posterior import SDEParameterPosterior
    B, T, S, P, dt = 37, 30, 2, 3, 0.1
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)
    H = rn(3, S) if with_matrix else None
    O = 3 if with_matrix else S
    obs = Observations(times=torch.tensor([0.0, 0.9, 1.0, 2.5, 3.0, 7.0]).to(DEV), values=(rn(6, O) + 1.0).to(DEV))  # 7.0 clamps to T
    lik = GaussianObservationLikelihood(variance=0.3, obs_matrix=None if H is None else H.to(DEV))
    prior = Prior(type=PriorType.LOG_NORMAL if prior_type == "log_normal" else PriorType.NORMAL, mean=0.2, std=1.3, dim=P)
    post = SDEParameterPosterior(P, theta_pos).to(DEV)
    with torch.no_grad():
        post.mean.copy_((rn(P) * 0.3).to(DEV)); post.log_std.copy_((rn(P) * 0.2 - 0.5).to(DEV))
    base = dict(z=rn(B, T + 1, S).cumsum(1) * 0.1 + 1.0, means=rn(B, T, S) * 0.3,
                chol=torch.tril(rn(B, T, S, S) * 0.2, -1) + torch.diag_embed(torch.rand(B, T, S, generator=g) + 0.5),
                theta=torch.rand(B, P, generator=g) * 0.8 + 0.1)
    results = {}
    for fused in (True, False):
        em.HIP_TAIL = fused
        try:
            leaves = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
            sample = DiffusionPathSample(z=leaves["z"], transition_means=leaves["means"], transition_cholesky=leaves["chol"],
                                         state_space=StateSpace(S, [0, 1]))
            res = em.compute_evidence_lower_bound(LotkaVolterra(), obs, lik, prior, post, leaves["theta"], sample, dt)
            c = res.components
            vals = [res.evidence_lower_bound, c.observation_log_prob, c.sde_log_prob, c.generative_log_prob, c.prior_log_prob,
                    c.posterior_log_prob]
            grads = torch.autograd.grad(res.evidence_lower_bound, list(leaves.values()) + [post.mean, post.log_std])
            results[fused] = ([float(v) for v in vals], [t.double().cpu().numpy() for t in grads])
        finally:
            em.HIP_TAIL = True
    for a, b_ in zip(*[results[f][0] for f in (True, False)]):
        assert abs(a - b_) <= 5e-6 * max(1.0, abs(b_))
    for a, b_, nm in zip(results[True][1], results[False][1], list(base) + ["q_mean", "q_log_std"]):
        assert rel_err(a, b_) < 2e-5, nm


def test_fused_tail_component_gradients():
    """The five component means are differentiable outputs too (the trainer detaches them, a user need not)."""
    from viforsdes_amd import _hip
    B, K, S, P = 9, 4, 2, 3
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(B, K, S, generator=g) + 0.5).to(DEV); y = torch.rand(K, S, generator=g).to(DEV)
    th = (torch.rand(B, P, generator=g) + 0.2).to(DEV)
    mean, ls = torch.zeros(P, device=DEV), torch.full((P,), -0.3, device=DEV)
    paths = [torch.randn(B, generator=g).to(DEV) for _ in range(3)]
    w = torch.tensor([0.0, 1.0, -2.0, 0.5, 3.0, -1.5], device=DEV)   # upstream gradient on the components only
    got = _hip.elbo_tail_bwd(x, y, None, 0.5, th, 1, 0.0, 1.0, mean, ls, [0, 1, 2], w)
    xr, tr, mr, lr = (t.clone().double().requires_grad_(True) for t in (x, th, mean, ls))
    obs = (-0.5 * (y.double() - xr) ** 2 / 0.5 - 0.5 * np.log(2 * np.pi * 0.5)).sum((1, 2))
    lg = tr.log()
    prior = (-0.5 * lg ** 2 - 0.5 * np.log(2 * np.pi) - lg).sum(-1)
    zq = (lg - mr) * torch.exp(-lr)
    post = (-0.5 * zq ** 2 - lr - 0.5 * np.log(2 * np.pi) - lg).sum(-1)
    loss = 1.0 * obs.mean() + 3.0 * prior.mean() - 1.5 * post.mean()
    ref = torch.autograd.grad(loss, [xr, tr, mr, lr])
    for a, b_ in zip(got[:4], ref):
        assert rel_err(a.cpu().numpy(), b_.cpu().numpy()) < 1e-5
    assert torch.allclose(got[4], torch.full((B,), -2.0 / B, device=DEV)) and torch.allclose(got[5], torch.full((B,), 0.5 / B, device=DEV))
    assert float(got[6].abs().max()) == 0.0
