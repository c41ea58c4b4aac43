"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference lives read-only at /root/reference and
never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is recorded (inputs and expected outputs only -- no reference source text):

* head_*.npz      fused-head forward/backward cases.  Expected values come from the
                  reference's own eager single-step definition
                  ``DiffusionTransitionHead.forward`` (models/head.py:68-97) looped over
                  time with the Euler-Maruyama update of kernels/forward.py:365, and its
                  autograd (this is oracle "O1" of SURVEY.md section 8c).  For the tiny cases the
                  reference's Triton kernels themselves are also run under
                  TRITON_INTERPRET=1 ("O2") and stored next to O1.
* elbo_*.npz      ``compute_evidence_lower_bound`` (inference/evidence_lower_bound.py:19-74)
                  with the example OU / Lotka-Volterra SDEs: per-sample terms, components,
                  scalar and autograd gradients.
* encoder_tiny.npz  ``ObservationContextEncoder`` forward + gradients for a tiny config.
* state_dict_manifest.json  key -> [shape, dtype] of ``VariationalSDEPosterior.state_dict()``.
* fused_dims.npz   encoder forward/gradients and a K=20-step trainer trajectory from one initial state at dims
                  where the build's fused encoder route is active (hidden 128, 2 heads, depth 2, batch 104).
* encoder_d128.npz  ``ObservationContextEncoder`` forward + gradients at head_dim 128 (hidden 128, one head, depth 2, batch 104).
* euler_maruyama.npz  ``euler_maruyama`` trajectories + gradients for the example OU / LV SDEs with injected noise.
* posterior_sample.npz  ``VariationalPosterior.sample`` / ``.summary`` under the EMA swap (live weights != shadow), draws recorded.
* pretrain.npz     ``pretrain_sde_parameters`` (OU with a free dimension, LV with one non-finite iteration), draws recorded.
* trajectory_tiny.npz  a K-step ``VariationalInferenceTrainer`` run on CPU with the head's
                  kernel call replaced by O1 and every ``torch.randn`` draw recorded.
"""
from __future__ import annotations

import json
import os
import sys
import typing

import numpy as np
import typing_extensions

typing.Self = typing_extensions.Self  # reference needs py>=3.11 (config.py:5)
os.environ.setdefault("TRITON_INTERPRET", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

from variational_sde.config import EncoderConfig, HeadConfig, TrainingConfig  # noqa: E402
from variational_sde.core.observations import GaussianObservationLikelihood, Observations  # noqa: E402
from variational_sde.core.priors import Prior, PriorType  # noqa: E402
from variational_sde.inference.evidence_lower_bound import compute_evidence_lower_bound  # noqa: E402
from variational_sde.inference.state_space import StateSpace  # noqa: E402
from variational_sde.inference.types import DiffusionPathSample  # noqa: E402
from variational_sde.models.head import DiffusionTransitionHead  # noqa: E402
from variational_sde.models.encoder import ObservationContextEncoder  # noqa: E402
from variational_sde.models.sde_parameter_posterior import SDEParameterPosterior  # noqa: E402
from variational_sde.models.variational_sde_posterior import VariationalSDEPosterior  # noqa: E402


def o1(head, z0, ctx, theta, eps, dt):
    """Loop the reference's eager step (head.py:68-86) with the update of forward.py:365."""
    h = head.init_hidden(z0.shape[0], z0.device, z0.dtype)
    z, P, M, Ls = z0, [z0], [], []
    for t in range(ctx.shape[1]):
        mu, L, h = head(z, ctx[:, t], theta, h)
        z = z + mu * dt + torch.einsum("bij,bj->bi", L, eps[:, t]) * dt ** 0.5
        P.append(z); M.append(mu); Ls.append(L)
    return torch.stack(P, 1), torch.stack(M, 1), torch.stack(Ls, 1)


def head_weights(head):
    ws = head._extract_gru_weights()
    return list(ws) + [head.out_proj.weight, head.out_proj.bias]


W_NAMES = ["W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0", "W_ih_stack", "W_hh_stack",
           "b_ih_stack", "b_hh_stack", "out_weight", "out_bias"]
G_NAMES = ["x0", "context", "sde_parameters"] + W_NAMES


def make_head_case(name, B, T, S, C, P, H, L, seed, with_o2, with_f64=True, clamp_stress=False):
    g = torch.Generator().manual_seed(seed)
    head = DiffusionTransitionHead(S, C, P, HeadConfig(hidden_dim=H, num_layers=L))
    with torch.no_grad():
        for p in head.gru.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.35)
        head.out_proj.weight.copy_(torch.randn(head.out_proj.weight.shape, generator=g) * 0.3)
        bias = torch.randn(head.out_proj.bias.shape, generator=g) * 0.1
        for k in range(S):
            # default init puts 1.0 on the diagonal entries (head.py:60-66); the stress case
            # straddles DIAG_MIN=0.01 so that both branches of the clamp gradient rule fire.
            bias[S + k * (k + 3) // 2] += 0.02 if clamp_stress else 0.6
        head.out_proj.bias.copy_(bias)
    dt = 0.05
    x0 = torch.randn(B, S, generator=g)
    ctx_full = torch.randn(B, T + 1, C, generator=g)
    theta = torch.randn(B, P, generator=g).abs() + 0.1
    eps = torch.randn(B, T, S, generator=g)
    gp = torch.randn(B, T + 1, S, generator=g)
    gm = torch.randn(B, T, S, generator=g)
    gl = torch.randn(B, T, S, S, generator=g)

    rec = {"dims": np.array([B, T, S, C, P, H, L]), "dt": np.array(dt), "x0": x0.numpy(),
           "context_full": ctx_full.numpy(), "sde_parameters": theta.numpy(), "eps": eps.numpy(),
           "g_paths": gp.numpy(), "g_means": gm.numpy(), "g_chol": gl.numpy()}
    for n, w in zip(W_NAMES, head_weights(head)):
        rec["w_" + n] = w.detach().numpy().copy()

    def run(dtype, tag):
        hd = head.double() if dtype == torch.float64 else head.float()
        ins = [t.to(dtype).clone().requires_grad_(True) for t in (x0, ctx_full, theta)]
        ws = head_weights(hd)
        paths, means, chol = o1(hd, ins[0], ins[1][:, :-1], ins[2], eps.to(dtype), dt)
        rec[f"{tag}_paths"] = paths.detach().numpy().copy()
        rec[f"{tag}_means"] = means.detach().numpy().copy()
        rec[f"{tag}_chol"] = chol.detach().numpy().copy()
        loss = (paths * gp.to(dtype)).sum() + (means * gm.to(dtype)).sum() + (chol * gl.to(dtype)).sum()
        params = [hd.gru.weight_ih_l0, hd.gru.weight_hh_l0, hd.gru.bias_ih_l0, hd.gru.bias_hh_l0]
        stacks = [[getattr(hd.gru, f"{k}_l{l}") for l in range(1, L)]
                  for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        flat = params + [p for st in stacks for p in st] + [hd.out_proj.weight, hd.out_proj.bias]
        grads = torch.autograd.grad(loss, ins + flat)
        gx0, gctx, gth = grads[:3]
        rec[f"{tag}_grad_x0"] = gx0.numpy().copy()
        rec[f"{tag}_grad_context"] = gctx[:, :-1].numpy().copy()  # grad wrt context[:, :-1]
        rec[f"{tag}_grad_sde_parameters"] = gth.numpy().copy()
        gi = 3
        for n in W_NAMES[:4]:
            rec[f"{tag}_grad_{n}"] = grads[gi].numpy().copy(); gi += 1
        for n in W_NAMES[4:8]:
            if L > 1:
                rec[f"{tag}_grad_{n}"] = torch.stack(list(grads[gi:gi + L - 1])).numpy().copy()
            else:
                rec[f"{tag}_grad_{n}"] = np.zeros((0,) + tuple(ws[W_NAMES.index(n)].shape[1:]))
            gi += L - 1
        rec[f"{tag}_grad_out_weight"] = grads[gi].numpy().copy()
        rec[f"{tag}_grad_out_bias"] = grads[gi + 1].numpy().copy()

    run(torch.float32, "o1f32")
    if with_f64:
        run(torch.float64, "o1f64")
    head.float()

    if with_o2:
        # The reference's Triton kernels under the interpreter.
        import triton.language as tl
        from triton.language.extra import libdevice
        libdevice.tanh = lambda x: 1.0 - 2.0 / (tl.exp(2.0 * x) + 1.0)
        import variational_sde.kernels.forward as kf
        kf.libdevice = libdevice
        head.train()
        ins = [t.clone().requires_grad_(True) for t in (x0, ctx_full, theta)]
        paths, means, chol = head.sample_diffusion_paths(ins[0], ins[1][:, :-1], ins[2], eps, dt)
        rec["o2_paths"] = paths.detach().numpy().copy()
        rec["o2_means"] = means.detach().numpy().copy()
        rec["o2_chol"] = chol.detach().numpy().copy()
        loss = (paths * gp).sum() + (means * gm).sum() + (chol * gl).sum()
        flat = [head.gru.weight_ih_l0, head.gru.weight_hh_l0, head.gru.bias_ih_l0, head.gru.bias_hh_l0,
                head.out_proj.weight, head.out_proj.bias]
        grads = torch.autograd.grad(loss, ins + flat)
        rec["o2_grad_x0"] = grads[0].numpy().copy()
        rec["o2_grad_context"] = grads[1][:, :-1].numpy().copy()
        rec["o2_grad_sde_parameters"] = grads[2].numpy().copy()
        rec["o2_grad_W_ih_l0"] = grads[3].numpy().copy()
        rec["o2_grad_W_hh_l0"] = grads[4].numpy().copy()
        rec["o2_grad_b_ih_l0"] = grads[5].numpy().copy()
        rec["o2_grad_b_hh_l0"] = grads[6].numpy().copy()
        rec["o2_grad_out_weight"] = grads[7].numpy().copy()
        rec["o2_grad_out_bias"] = grads[8].numpy().copy()
        d = max(float(np.abs(rec["o2_" + k] - rec["o1f32_" + k]).max()) for k in
                ("paths", "means", "chol", "grad_x0", "grad_context", "grad_W_ih_l0", "grad_out_weight"))
        print(f"  {name}: max |O2 - O1| = {d:.3e}")
    np.savez_compressed(os.path.join(OUT, f"head_{name}.npz"), **rec)
    print("wrote", f"head_{name}.npz")


# ----------------------------------------------------------------------------- ELBO
def example_sdes():
    from examples.ornstein_uhlenbeck import OrnsteinUhlenbeck
    from examples.lotka_volterra import LotkaVolterra
    return OrnsteinUhlenbeck(), LotkaVolterra()


def make_elbo_case(name, sde, obs_times, obs_values, var, prior, state_pos, theta_pos, B, T, dt, seed):
    g = torch.Generator().manual_seed(seed)
    S, P = sde.state_dim, sde.sde_param_dim
    space = StateSpace(S, state_pos)
    post = SDEParameterPosterior(P, theta_pos)
    with torch.no_grad():
        post.mean.copy_(torch.randn(P, generator=g) * 0.3)
        post.log_std.copy_(torch.randn(P, generator=g) * 0.2 - 0.5)
    eps_theta = torch.randn(B, P, generator=g)
    std = post.log_std.exp()
    theta = post.mean + std * eps_theta
    theta = torch.where(post.positive_mask, theta.exp(), theta).detach().requires_grad_(True)
    observations = Observations(times=torch.tensor(obs_times), values=torch.tensor(obs_values))
    x0 = observations.values[0].expand(B, -1)
    z0 = space.to_latent(x0)
    # a plausible random path in latent space + random transition params
    incr = torch.randn(B, T, S, generator=g) * (dt ** 0.5) * 0.8
    z = torch.cat([z0[:, None], z0[:, None] + incr.cumsum(1)], 1).detach().requires_grad_(True)
    means = (torch.randn(B, T, S, generator=g) * 0.5).requires_grad_(True)
    Lm = torch.randn(B, T, S, S, generator=g) * 0.3
    Lm = torch.tril(Lm, -1) + torch.diag_embed(torch.rand(B, T, S, generator=g) + 0.3)
    chol = Lm.detach().requires_grad_(True)
    like = GaussianObservationLikelihood(variance=var)
    sample = DiffusionPathSample(z=z, transition_means=means, transition_cholesky=chol, state_space=space)

    # per-sample terms, recomputed exactly as evidence_lower_bound.py:29-61 does
    res = compute_evidence_lower_bound(sde, observations, like, prior, post, theta, sample, dt)
    grads = torch.autograd.grad(res.evidence_lower_bound, [z, means, chol, theta, post.mean, post.log_std])
    from variational_sde.inference.evidence_lower_bound import _gaussian_log_prob
    from einops import rearrange, repeat
    with torch.no_grad():
        x = sample.x
        xt = rearrange(x[:, :-1], "b t d -> (b t) d")
        th = repeat(theta, "b d -> (b t) d", t=T)
        drift = rearrange(sde.drift(xt, th), "(b t) d -> b t d", b=B)
        diff = rearrange(sde.diffusion(xt, th), "(b t) d e -> b t d e", b=B)
        sde_lp = _gaussian_log_prob(x[:, 1:], x[:, :-1] + drift * dt, diff * dt ** 0.5)
        gen_lp = _gaussian_log_prob(z[:, 1:], z[:, :-1] + means * dt, chol * dt ** 0.5)
        jac = sample.log_jacobian()
        idx = torch.clamp(torch.round(observations.times / dt).long(), max=T)
        obs_lp = like.log_prob(repeat(observations.values, "t d -> b t d", b=B), x[:, idx]).sum(-1)
        prior_lp = prior.log_prob(theta)
        post_lp = post.log_prob(theta)
    c = res.components
    rec = dict(
        dims=np.array([B, T, S, P]), dt=np.array(dt), variance=np.array(var),
        state_positive_dims=np.array(state_pos, dtype=np.int64), theta_positive_dims=np.array(theta_pos, dtype=np.int64),
        prior_type=np.array(1 if prior.type == PriorType.LOG_NORMAL else 0), prior_mean=np.array(prior.mean),
        prior_std=np.array(prior.std), obs_times=np.array(obs_times, dtype=np.float32),
        obs_values=np.array(obs_values, dtype=np.float32), obs_idx=idx,
        q_mean=post.mean, q_log_std=post.log_std, eps_theta=eps_theta,
        theta=theta, z=z, x=x, z0=z0, x0=x0,
        means=means, chol=chol, drift=drift, diffusion=diff,
        sde_lp=sde_lp, gen_lp=gen_lp, jac=jac, obs_lp=obs_lp,
        prior_lp=prior_lp, post_lp=post_lp, elbo=res.evidence_lower_bound,
        comp_obs=c.observation_log_prob, comp_sde=c.sde_log_prob,
        comp_gen=c.generative_log_prob, comp_prior=c.prior_log_prob,
        comp_post=c.posterior_log_prob, expected_value=post.expected_value,
        grad_z=grads[0], grad_means=grads[1], grad_chol=grads[2],
        grad_theta=grads[3], grad_q_mean=grads[4], grad_q_log_std=grads[5])
    rec = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(OUT, f"elbo_{name}.npz"), **rec)
    print("wrote", f"elbo_{name}.npz", "elbo =", float(res.evidence_lower_bound))


# -------------------------------------------------------------------------- encoder
def randomize_(module, gen, scale=0.05):
    """Give zero-initialised modulators/gates non-trivial values so the test sees them."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if not p.requires_grad:
                continue
            if p.abs().sum() == 0 or "v_residual_lambda" in n:
                p.add_(torch.randn(p.shape, generator=gen) * scale)


def make_encoder_case():
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    cfg = EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=3, mlp_ratio=8 / 3)
    enc = ObservationContextEncoder(observation_dim=2, sde_param_dim=3, config=cfg)
    randomize_(enc, g, 0.2)
    obs_t = torch.tensor([0.0, 0.5, 1.0, 1.5])
    obs_v = torch.randn(4, 2, generator=g)
    theta = (torch.randn(5, 3, generator=g).abs() + 0.2).requires_grad_(True)
    ctx = enc(obs_v, obs_t, theta, 1.5, 0.1)
    gout = torch.randn(ctx.shape, generator=g)
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    params = [p for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx * gout).sum(), [theta] + params)
    rec = {"obs_times": obs_t.numpy(), "obs_values": obs_v.numpy(), "theta": theta.detach().numpy(),
           "time_horizon": np.array(1.5), "time_step": np.array(0.1), "context": ctx.detach().numpy(),
           "g_context": gout.numpy(), "grad_theta": grads[0].numpy(),
           "cfg": np.array([cfg.hidden_dim, cfg.cond_dim, cfg.num_heads, cfg.depth])}
    for k, v in enc.state_dict().items():
        rec["sd::" + k] = torch.view_as_real(v).numpy() if v.is_complex() else v.numpy()
    for n, gr in zip(names, grads[1:]):
        rec["grad::" + n] = gr.numpy()
    np.savez_compressed(os.path.join(OUT, "encoder_tiny.npz"), **rec)
    print("wrote encoder_tiny.npz", tuple(ctx.shape))


def make_manifest():
    out = {}
    for depth in (2, 8):
        m = VariationalSDEPosterior(2, 2, 3, EncoderConfig(hidden_dim=256, num_heads=4, depth=depth),
                                    HeadConfig(hidden_dim=64, num_layers=2), [0, 1, 2])
        out[f"lv_depth{depth}"] = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
        out[f"lv_depth{depth}_params"] = [n for n, _ in m.named_parameters()]
    m = VariationalSDEPosterior(1, 1, 3, EncoderConfig(), HeadConfig(hidden_dim=32, num_layers=3), [0, 2])
    out["ou_default_l3"] = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote state_dict_manifest.json")


# ----------------------------------------------------------------------- trajectory
def make_trajectory():
    """K optimizer steps of the reference trainer on CPU (mixed precision off, compile off)
    with the Triton call replaced by O1 and every torch.randn draw recorded in order."""
    from variational_sde.inference.trainer import VariationalInferenceTrainer
    from variational_sde.console import Console
    _, lv = example_sdes()
    K, B, dt, horizon = 12, 6, 0.1, 1.0
    obs = Observations(times=torch.tensor([0.0, 0.5, 1.0]),
                       values=torch.tensor([[1.2, 0.7], [0.9, 1.1], [0.6, 1.4]]))
    prior = Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3)

    def patched(self, x0, context, sde_parameters, standard_noise, time_step):
        return o1(self, x0, context, sde_parameters, standard_noise, time_step)

    DiffusionTransitionHead.sample_diffusion_paths = patched
    torch.manual_seed(2024)
    tr = VariationalInferenceTrainer(
        sde=lv, observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=0.25),
        prior=prior, time_horizon=horizon,
        config=TrainingConfig(time_step=dt, batch_size=B, n_iterations=K, learning_rate=1e-3, sde_param_lr=1e-2),
        encoder_config=EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=2),
        head_config=HeadConfig(hidden_dim=16, num_layers=2),
        state_positive_dims=[0, 1], sde_param_positive_dims=[0, 1, 2], device="cpu",
        mixed_precision=False, console=Console(enabled=False), accelerator=None)
    model = tr.ctx.model
    g = torch.Generator().manual_seed(5)
    randomize_(model.encoder, g, 0.1)
    with torch.no_grad():
        model.head.out_proj.weight.add_(torch.randn(model.head.out_proj.weight.shape, generator=g) * 0.2)
    tr.ctx.ema._init_shadow()
    rec = {}
    for k, v in model.state_dict().items():
        rec["init::" + k] = torch.view_as_real(v).numpy().copy() if v.is_complex() else v.numpy().copy()
    draws = []
    real_randn = torch.randn

    def rec_randn(*a, **kw):
        t = real_randn(*a, **kw)
        draws.append(t.detach().clone())
        return t

    torch.randn = rec_randn
    elbos, comps, gnorms = [], [], []
    try:
        model.train()
        for step in range(K):
            r = tr._train_step(model)
            tr.ctx.ema.update()
            elbos.append(r.elbo_result.evidence_lower_bound.item())
            c = r.elbo_result.components
            comps.append([c.observation_log_prob.item(), c.sde_log_prob.item(), c.generative_log_prob.item(),
                          c.prior_log_prob.item(), c.posterior_log_prob.item()])
            gnorms.append(r.grad_norm)
    finally:
        torch.randn = real_randn
    assert len(draws) == 2 * K
    rec["theta_eps"] = torch.stack(draws[0::2]).numpy()
    rec["path_noise"] = torch.stack(draws[1::2]).numpy()
    rec["elbo"] = np.array(elbos); rec["components"] = np.array(comps); rec["grad_norm"] = np.array(gnorms)
    post = model.sde_parameter_posterior
    rec["final_mean"] = post.mean.detach().numpy().copy()
    rec["final_log_std"] = post.log_std.detach().numpy().copy()
    rec["final_expected_value"] = post.expected_value.detach().numpy().copy()
    rec["ema_mean"] = tr.ctx.ema.shadow["sde_parameter_posterior.mean"].numpy().copy()
    rec["ema_log_std"] = tr.ctx.ema.shadow["sde_parameter_posterior.log_std"].numpy().copy()
    rec["final::head.out_proj.bias"] = model.head.out_proj.bias.detach().numpy().copy()
    rec["final::encoder.bridge_token"] = model.encoder.bridge_token.detach().numpy().copy()
    rec["cfg"] = np.array([K, B]); rec["dt"] = np.array(dt); rec["horizon"] = np.array(horizon)
    rec["obs_times"] = obs.times.numpy(); rec["obs_values"] = obs.values.numpy()
    np.savez_compressed(os.path.join(OUT, "trajectory_tiny.npz"), **rec)
    print("wrote trajectory_tiny.npz; elbo[0], elbo[-1] =", elbos[0], elbos[-1])


# --------------------------------------------------- fused-eligible dims: encoder case + trajectory from ONE init
def make_fused_dims():
    """Encoder hidden 128 / 2 heads (head_dim 64) / depth 2, head GRU 64 x 2, LV, 41 grid tokens, batch 104
    (104 * 41 = 4264 token rows >= the 4096-row threshold of the packed bf16 GEMM / weight-gradient kernels): the
    dims at which the build's fused encoder route, its own attention kernels and its packed Linears are all active.
    One initial ``state_dict`` serves (a) an encoder forward/gradient case and (b) a K=20-step trainer trajectory
    (reference CPU path, fp32, O1 head, every ``torch.randn`` draw recorded).  The upstream context gradient is
    ``RandomState(seed).randn`` (a frozen stream), regenerated by the test instead of being stored."""
    from variational_sde.inference.trainer import VariationalInferenceTrainer
    from variational_sde.console import Console
    _, lv = example_sdes()
    K, B, dt, horizon = 20, 104, 0.05, 2.0
    obs = Observations(times=torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0]),
                       values=torch.tensor([[1.2, 0.7], [0.9, 1.1], [0.6, 1.4], [0.8, 1.0], [1.1, 0.8]]))
    prior = Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3)

    def patched(self, x0, context, sde_parameters, standard_noise, time_step):
        return o1(self, x0, context, sde_parameters, standard_noise, time_step)

    DiffusionTransitionHead.sample_diffusion_paths = patched
    torch.manual_seed(4242)
    tr = VariationalInferenceTrainer(
        sde=lv, observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=0.25),
        prior=prior, time_horizon=horizon,
        config=TrainingConfig(time_step=dt, batch_size=B, n_iterations=K, learning_rate=1e-3, sde_param_lr=1e-2),
        encoder_config=EncoderConfig(hidden_dim=128, cond_dim=16, num_heads=2, depth=2),
        head_config=HeadConfig(hidden_dim=64, num_layers=2),
        state_positive_dims=[0, 1], sde_param_positive_dims=[0, 1, 2], device="cpu",
        mixed_precision=False, console=Console(enabled=False), accelerator=None)
    model = tr.ctx.model
    g = torch.Generator().manual_seed(9)
    randomize_(model.encoder, g, 0.1)
    with torch.no_grad():
        model.head.out_proj.weight.add_(torch.randn(model.head.out_proj.weight.shape, generator=g) * 0.2)
    tr.ctx.ema._init_shadow()
    rec = {}
    for k, v in model.state_dict().items():
        rec["init::" + k] = torch.view_as_real(v).numpy().copy() if v.is_complex() else v.numpy().copy()

    # (a) encoder forward + gradients at the initial weights
    enc = model.encoder
    theta = (torch.randn(B, 3, generator=g).abs() + 0.2).requires_grad_(True)
    ctx = enc(obs.values, obs.times, theta, horizon, dt)
    seed_g = 31337
    gout = torch.from_numpy(np.random.RandomState(seed_g).randn(*ctx.shape).astype(np.float32))
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    params = [p for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx * gout).sum(), [theta] + params)
    rec["enc_theta"] = theta.detach().numpy().copy()
    rec["enc_context_rows"] = np.arange(0, B, 13)                    # stored rows of the context (the rest is covered
    rec["enc_context"] = ctx.detach().numpy()[::13].copy()           # through the gradients of the shared weights)
    rec["enc_g_context_seed"] = np.array(seed_g)
    rec["enc_grad_theta"] = grads[0].numpy().copy()
    for n, gr in zip(names, grads[1:]):
        rec["enc_grad::" + n] = gr.numpy().copy()

    # (b) K-step trajectory
    draws = []
    real_randn = torch.randn

    def rec_randn(*a, **kw):
        t = real_randn(*a, **kw)
        draws.append(t.detach().clone())
        return t

    torch.randn = rec_randn
    elbos, comps, gnorms = [], [], []
    try:
        model.train()
        for step in range(K):
            r = tr._train_step(model)
            tr.ctx.ema.update()
            elbos.append(r.elbo_result.evidence_lower_bound.item())
            c = r.elbo_result.components
            comps.append([c.observation_log_prob.item(), c.sde_log_prob.item(), c.generative_log_prob.item(),
                          c.prior_log_prob.item(), c.posterior_log_prob.item()])
            gnorms.append(r.grad_norm)
    finally:
        torch.randn = real_randn
    assert len(draws) == 2 * K
    rec["theta_eps"] = torch.stack(draws[0::2]).numpy()
    rec["path_noise"] = torch.stack(draws[1::2]).numpy().astype(np.float32)
    rec["elbo"] = np.array(elbos); rec["components"] = np.array(comps); rec["grad_norm"] = np.array(gnorms)
    post = model.sde_parameter_posterior
    rec["final_mean"] = post.mean.detach().numpy().copy()
    rec["final_log_std"] = post.log_std.detach().numpy().copy()
    rec["final_expected_value"] = post.expected_value.detach().numpy().copy()
    rec["ema_mean"] = tr.ctx.ema.shadow["sde_parameter_posterior.mean"].numpy().copy()
    rec["cfg"] = np.array([K, B]); rec["dt"] = np.array(dt); rec["horizon"] = np.array(horizon)
    rec["obs_times"] = obs.times.numpy(); rec["obs_values"] = obs.values.numpy()
    np.savez_compressed(os.path.join(OUT, "fused_dims.npz"), **rec)
    print("wrote fused_dims.npz; elbo[0], elbo[-1] =", elbos[0], elbos[-1], "grad_norm", gnorms[0], gnorms[-1])


def make_encoder_d128():
    """``ObservationContextEncoder`` forward + gradients at head_dim 128 (hidden 128 / ONE head / depth 2, cond 16; 41 grid
    tokens x batch 104 = 4264 token rows): the dims at which the build's streamed attention kernels (D = 128), the QK-norm /
    RoPE kernels with 64 rotary pairs, the 128-wide gate and the packed Linears run -- the head_dim of BASELINE config 5
    (encoder 512 / 4 heads).  Depth 2 so that block 1 mixes the residual values of block 0.  The upstream context gradient is
    ``RandomState(seed).randn`` (a frozen stream), regenerated by the test instead of being stored."""
    g = torch.Generator().manual_seed(128)
    torch.manual_seed(128)
    cfg = EncoderConfig(hidden_dim=128, cond_dim=16, num_heads=1, depth=2)
    enc = ObservationContextEncoder(observation_dim=2, sde_param_dim=3, config=cfg)
    randomize_(enc, g, 0.1)
    B, dt, horizon = 104, 0.05, 2.0
    obs_t = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0])
    obs_v = torch.tensor([[1.2, 0.7], [0.9, 1.1], [0.6, 1.4], [0.8, 1.0], [1.1, 0.8]])
    theta = (torch.randn(B, 3, generator=g).abs() + 0.2).requires_grad_(True)
    ctx = enc(obs_v, obs_t, theta, horizon, dt)
    seed_g = 271828
    gout = torch.from_numpy(np.random.RandomState(seed_g).randn(*ctx.shape).astype(np.float32))
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    params = [p for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx * gout).sum(), [theta] + params)
    rec = {"obs_times": obs_t.numpy(), "obs_values": obs_v.numpy(), "theta": theta.detach().numpy(),
           "time_horizon": np.array(horizon), "time_step": np.array(dt),
           "context_rows": np.arange(0, B, 13), "context": ctx.detach().numpy()[::13].copy(),
           "g_context_seed": np.array(seed_g), "grad_theta": grads[0].numpy(),
           "cfg": np.array([cfg.hidden_dim, cfg.cond_dim, cfg.num_heads, cfg.depth, B])}
    for k, v in enc.state_dict().items():
        rec["sd::" + k] = torch.view_as_real(v).numpy() if v.is_complex() else v.numpy()
    for n, gr in zip(names, grads[1:]):
        rec["grad::" + n] = gr.numpy()
    np.savez_compressed(os.path.join(OUT, "encoder_d128.npz"), **rec)
    print("wrote encoder_d128.npz", tuple(ctx.shape))


# ------------------------------------------------------------------- Euler-Maruyama simulator of the model SDE
def make_em_cases():
    """``euler_maruyama`` (core/euler_maruyama.py:11-45) with the example OU / LV SDEs, injected noise: trajectory and the
    gradients of <trajectory, g> with respect to theta and x0 (what ``pretrain_sde_parameters`` differentiates,
    trainer.py:208-259).  The LV case starts two rows next to zero so that the positive-dims clamp (1e-6) fires."""
    from variational_sde.core.euler_maruyama import euler_maruyama
    ou, lv = example_sdes()
    rec = {}
    for name, sde, B, horizon, dt, pos in (("ou", ou, 5, 5.0, 0.05, []), ("lv", lv, 6, 4.0, 0.1, [0, 1])):
        g = torch.Generator().manual_seed(500 + len(name) + B)
        S, P = sde.state_dim, sde.sde_param_dim
        T = round(horizon / dt)
        theta = (torch.rand(B, P, generator=g) * 0.8 + 0.1).requires_grad_(True)
        x0 = (torch.rand(B, S, generator=g) * 2.0 + 0.5)
        if name == "lv":
            x0[0] = torch.tensor([2e-3, 1e-3]); x0[1] = torch.tensor([5e-3, 3.0])
        x0.requires_grad_(True)
        noise = torch.randn(B, T, S, generator=g)
        traj = euler_maruyama(sde, x0, theta, horizon, dt, pos, noise=noise)
        gw = torch.randn(traj.shape, generator=g)
        gth, gx0 = torch.autograd.grad((traj * gw).sum(), [theta, x0])
        n_clamped = int((traj[:, 1:][..., pos] == 1e-6).sum()) if pos else 0
        print(f"  em_{name}: T={T}, clamped entries {n_clamped}, |traj|max {float(traj.abs().max()):.3g}")
        for k, v in dict(theta=theta, x0=x0, noise=noise, traj=traj, g_traj=gw, grad_theta=gth, grad_x0=gx0).items():
            rec[f"{name}_{k}"] = v.detach().numpy().copy()
        rec[f"{name}_cfg"] = np.array([horizon, dt]); rec[f"{name}_pos"] = np.array(pos, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "euler_maruyama.npz"), **rec)
    print("wrote euler_maruyama.npz")


def make_sde_coeffs():
    """Drift / diffusion of the example OU and LV SDEs on the grid points of a few paths, evaluated exactly like
    inference/evidence_lower_bound.py:37-40 (flattened states, theta repeated over time), and torch autograd's gradients of
    <drift, g_f> + <diffusion, g_G>.  The LV paths contain near-zero states so that all three clamp(min=1e-6) fire."""
    ou, lv = example_sdes()
    rec = {}
    for name, sde, B, T in (("ou", ou, 4, 9), ("lv", lv, 6, 11)):
        g = torch.Generator().manual_seed(900 + B + T)
        S, P = sde.state_dim, sde.sde_param_dim
        theta = (torch.rand(B, P, generator=g) * 0.8 + 0.1).requires_grad_(True)
        x = torch.rand(B, T + 1, S, generator=g) * 3.0 + 0.2
        if name == "lv":
            x[0, :4] = torch.tensor([[1e-7, 2.0], [3.0, 1e-8], [1e-9, 1e-9], [1e-7, 1e-2]])
            x[1, 2] = torch.tensor([0.0, 0.0])
        x.requires_grad_(True)
        x_flat = x[:, :-1].reshape(B * T, S)
        theta_flat = theta.unsqueeze(1).expand(B, T, -1).reshape(B * T, -1)
        drift = sde.drift(x_flat, theta_flat).reshape(B, T, S)
        diffusion = sde.diffusion(x_flat, theta_flat).reshape(B, T, S, S)
        gf = torch.randn(drift.shape, generator=g); gG = torch.randn(diffusion.shape, generator=g)
        gx, gth = torch.autograd.grad((drift * gf).sum() + (diffusion * gG).sum(), [x, theta])
        for k, v in dict(x=x, theta=theta, drift=drift, diffusion=diffusion, g_drift=gf, g_diffusion=gG, grad_x=gx,
                         grad_theta=gth).items():
            rec[f"{name}_{k}"] = v.detach().numpy().copy()
        print(f"  sde_coeffs_{name}: B={B} T={T}, min diffusion diag {float(diffusion.diagonal(dim1=-2, dim2=-1).min()):.3g}")
    np.savez_compressed(os.path.join(OUT, "sde_coefficients.npz"), **rec)
    print("wrote sde_coefficients.npz")


# ------------------------------------------------ eval-side caller: VariationalPosterior.sample / summary under the EMA swap
def _record_randn():
    draws, real = [], torch.randn

    def rec(*a, **kw):
        t = real(*a, **kw)
        draws.append(t.detach().clone())
        return t
    return draws, real, rec


def make_posterior_sample():
    """``VariationalPosterior.sample(n)`` and ``.summary(n)`` of the reference (posterior/variational_posterior.py:93-144) on a
    model whose EMA shadow differs from its live weights, with the head's kernel call replaced by O1 and every ``torch.randn``
    draw recorded: {live state_dict, EMA shadow, draws} -> {theta, diffusion_paths, summary statistics}."""
    from variational_sde.inference.exponential_moving_average import ExponentialMovingAverage
    from variational_sde.posterior.variational_posterior import VariationalPosterior

    def patched(self, x0, context, sde_parameters, standard_noise, time_step):
        return o1(self, x0, context, sde_parameters, standard_noise, time_step)
    DiffusionTransitionHead.sample_diffusion_paths = patched
    rec = {}
    cases = {
        "lv": dict(S=2, P=3, state_pos=[0, 1], theta_pos=[0, 1, 2], dt=0.1, horizon=1.0,
                   times=[0.0, 0.5, 1.0], values=[[1.2, 0.7], [0.9, 1.1], [0.6, 1.4]],
                   prior=Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3), n=7, n_summary=48, seed=31),
        "ou": dict(S=1, P=3, state_pos=[], theta_pos=[0, 2], dt=0.05, horizon=1.0,
                   times=[0.0, 0.5, 1.0], values=[[2.0], [1.5], [0.8]],
                   prior=Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3), n=5, n_summary=40, seed=32),
    }
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        model = VariationalSDEPosterior(c["S"], c["S"], c["P"], EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=2),
                                        HeadConfig(hidden_dim=16, num_layers=2), c["theta_pos"])
        g = torch.Generator().manual_seed(c["seed"] + 100)
        randomize_(model.encoder, g, 0.1)
        with torch.no_grad():
            model.head.out_proj.weight.add_(torch.randn(model.head.out_proj.weight.shape, generator=g) * 0.2)
            model.sde_parameter_posterior.mean.add_(torch.randn(c["P"], generator=g) * 0.3)
            model.sde_parameter_posterior.log_std.add_(torch.randn(c["P"], generator=g) * 0.2 - 1.0)
        ema = ExponentialMovingAverage(model)
        with torch.no_grad():  # the averaged weights are NOT the live ones: sample() must run on the shadow
            for k, v in ema.shadow.items():
                v.add_(torch.randn(v.shape, generator=g) * 0.05 * (v.abs().mean() + 0.05))
        obs = Observations(times=torch.tensor(c["times"]), values=torch.tensor(c["values"]))
        vp = VariationalPosterior(model=model, exponential_moving_average=ema, prior=c["prior"], observations=obs,
                                  time_horizon=c["horizon"], time_step=c["dt"], state_space=StateSpace(c["S"], c["state_pos"]),
                                  evidence_lower_bound_history=[-3.0, -2.0], device=torch.device("cpu"))
        for k, v in model.state_dict().items():
            rec[f"{name}::init::{k}"] = torch.view_as_real(v).numpy().copy() if v.is_complex() else v.numpy().copy()
        for k, v in ema.shadow.items():
            rec[f"{name}::ema::{k}"] = v.numpy().copy()
        draws, real, recfn = _record_randn()
        torch.randn = recfn
        try:
            s = vp.sample(c["n"])
            summ = vp.summary(c["n_summary"])
        finally:
            torch.randn = real
        assert len(draws) == 4 and draws[0].shape == (c["n"], c["P"])
        for k, v in model.state_dict().items():  # the swap was undone
            ref = rec[f"{name}::init::{k}"]
            assert np.array_equal(torch.view_as_real(v).numpy() if v.is_complex() else v.numpy(), ref)
        q = summ.sde_parameter_quantiles
        rec.update({f"{name}::sample_theta_eps": draws[0].numpy(), f"{name}::sample_noise": draws[1].numpy(),
                    f"{name}::summary_theta_eps": draws[2].numpy(), f"{name}::summary_noise": draws[3].numpy(),
                    f"{name}::sde_parameters": s.sde_parameters.numpy(), f"{name}::diffusion_paths": s.diffusion_paths.numpy(),
                    f"{name}::summary_mean": summ.sde_parameter_mean.numpy(), f"{name}::summary_std": summ.sde_parameter_std.numpy(),
                    f"{name}::summary_quantiles": torch.stack([q.q05, q.q25, q.q50, q.q75, q.q95]).numpy(),
                    f"{name}::summary_path_mean": summ.diffusion_path_mean.numpy(),
                    f"{name}::summary_path_std": summ.diffusion_path_std.numpy(),
                    f"{name}::obs_times": obs.times.numpy(), f"{name}::obs_values": obs.values.numpy(),
                    f"{name}::cfg": np.array([c["S"], c["P"], c["n"], c["n_summary"]]),
                    f"{name}::state_pos": np.array(c["state_pos"], dtype=np.int64),
                    f"{name}::theta_pos": np.array(c["theta_pos"], dtype=np.int64),
                    f"{name}::dt": np.array(c["dt"]), f"{name}::horizon": np.array(c["horizon"])})
        d = vp.diagnostics()
        rec[f"{name}::diagnostics"] = np.array([d.final_evidence_lower_bound, d.n_iterations])
        print(f"  posterior_sample {name}: theta[0] {s.sde_parameters[0].tolist()}, path mean {float(s.diffusion_paths.mean()):.4f}")
    np.savez_compressed(os.path.join(OUT, "posterior_sample.npz"), **rec)
    print("wrote posterior_sample.npz")


# ------------------------------------------------------------------ pre-training loop (trainer.py:208-259) with recorded draws
def make_pretrain():
    """``VariationalInferenceTrainer.pretrain_sde_parameters`` of the reference on CPU, every ``torch.randn`` draw recorded (the
    free-dimension init, then per iteration the theta noise [B, P] and the path noise [B, T, S]); per-iteration loss, best loss,
    median sigma as the reference reports them to its progress sink, and the returned mean.  One iteration's theta noise is
    scaled up so that the loss is non-finite there: the reference then skips that update (trainer.py:243-246)."""
    from contextlib import contextmanager
    from variational_sde.config import PretrainConfig
    from variational_sde.console import Console
    from variational_sde.inference.trainer import VariationalInferenceTrainer
    ou, lv = example_sdes()
    rec = {}
    cases = {
        "ou": dict(sde=ou, state_pos=[], theta_pos=[0, 2], dt=0.05, horizon=2.0, times=[0.0, 1.0, 2.0],
                   values=[[2.0], [1.5], [0.8]], prior=Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3),
                   var=0.1, K=10, B=48, bad_step=None, seed=41),
        "lv": dict(sde=lv, state_pos=[0, 1], theta_pos=[0, 1, 2], dt=0.1, horizon=2.0, times=[0.0, 1.0, 2.0],
                   values=[[1.2, 0.7], [0.9, 1.1], [0.6, 1.4]], prior=Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3),
                   var=0.25, K=10, B=48, bad_step=4, seed=42),
    }
    for name, c in cases.items():
        log = []

        class Sink:
            def update(self, step, mse, best, sigma_median):
                log.append([float(mse), float(best), float(sigma_median)])

        class RecConsole(Console):
            @contextmanager
            def pretrain_progress(self, n_iterations):
                yield Sink()

        torch.manual_seed(c["seed"])
        obs = Observations(times=torch.tensor(c["times"]), values=torch.tensor(c["values"]))
        tr = VariationalInferenceTrainer(
            sde=c["sde"], observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=c["var"]),
            prior=c["prior"], time_horizon=c["horizon"],
            config=TrainingConfig(time_step=c["dt"], batch_size=4, n_iterations=2),
            encoder_config=EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=1),
            head_config=HeadConfig(hidden_dim=16, num_layers=1), state_positive_dims=c["state_pos"],
            sde_param_positive_dims=c["theta_pos"], device="cpu", mixed_precision=False, console=RecConsole(enabled=False),
            accelerator=None)
        draws, real = [], torch.randn
        P = c["sde"].sde_param_dim
        n_free = P - len(c["theta_pos"])

        def rec_randn(*a, **kw):
            t = real(*a, **kw)
            k_theta = (len(draws) - (1 if n_free else 0))
            if c["bad_step"] is not None and tuple(t.shape) == (c["B"], P) and k_theta // 2 == c["bad_step"]:
                t = t * 200.0  # exp(mu + sigma * 200 eps) overflows: a non-finite loss, the update must be skipped
            draws.append(t.detach().clone())
            return t
        torch.randn = rec_randn
        try:
            best_mu = tr.pretrain_sde_parameters(PretrainConfig(n_iterations=c["K"], batch_size=c["B"], learning_rate=0.02))
        finally:
            torch.randn = real
        off = 1 if n_free else 0
        assert len(draws) == off + 2 * c["K"], (len(draws), off)
        if off:
            rec[f"{name}::init_draw"] = draws[0].numpy()
        rec[f"{name}::theta_eps"] = torch.stack(draws[off::2]).numpy()
        rec[f"{name}::path_noise"] = torch.stack(draws[off + 1::2]).numpy()
        rec[f"{name}::log"] = np.array(log)
        rec[f"{name}::best_mu"] = best_mu.numpy()
        rec[f"{name}::obs_times"] = obs.times.numpy(); rec[f"{name}::obs_values"] = obs.values.numpy()
        rec[f"{name}::cfg"] = np.array([c["K"], c["B"], -1 if c["bad_step"] is None else c["bad_step"]])
        rec[f"{name}::dt"] = np.array(c["dt"]); rec[f"{name}::horizon"] = np.array(c["horizon"]); rec[f"{name}::var"] = np.array(c["var"])
        rec[f"{name}::state_pos"] = np.array(c["state_pos"], dtype=np.int64)
        rec[f"{name}::theta_pos"] = np.array(c["theta_pos"], dtype=np.int64)
        nonfinite = [i for i, r in enumerate(log) if not np.isfinite(r[0])]
        print(f"  pretrain {name}: mse {log[0][0]:.4g} -> {log[-1][0]:.4g}, best {log[-1][1]:.4g}, non-finite steps {nonfinite}, "
              f"best_mu {best_mu.tolist()}")
    np.savez_compressed(os.path.join(OUT, "pretrain.npz"), **rec)
    print("wrote pretrain.npz")


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"head", "elbo", "encoder", "manifest", "trajectory"}
    if "head" in which:
        make_head_case("tiny_l2", 4, 7, 2, 16, 3, 8, 2, seed=11, with_o2=True)
        make_head_case("tiny_l1_odd", 3, 5, 3, 10, 2, 12, 1, seed=12, with_o2=True)
        make_head_case("tiny_l4", 2, 4, 1, 6, 3, 16, 4, seed=13, with_o2=False)
        make_head_case("s8_h64", 2, 6, 8, 16, 8, 64, 2, seed=14, with_o2=False)
        make_head_case("clamp", 4, 9, 3, 8, 2, 24, 2, seed=15, with_o2=False, clamp_stress=True)
        make_head_case("ou_dims", 3, 10, 1, 64, 3, 64, 2, seed=16, with_o2=False)
        make_head_case("lv_dims", 2, 12, 2, 256, 3, 64, 2, seed=17, with_o2=False, with_f64=False)
    if "head_wide" in which:
        make_head_case("h80_l2", 2, 6, 2, 8, 2, 80, 2, seed=18, with_o2=False)
        make_head_case("h130_l1_s10", 2, 4, 10, 6, 2, 130, 1, seed=19, with_o2=False)
        make_head_case("h96_l3", 2, 5, 3, 8, 3, 96, 3, seed=20, with_o2=False)
    if "head_mp" in which:  # hidden_dim 64 / two layers / state_dim <= 2 at batches where the dispatcher takes the multi-path MFMA kernels
        make_head_case("mp_b300_s2", 300, 10, 2, 32, 3, 64, 2, seed=27, with_o2=False)      # forward multi-path, backward v2
        make_head_case("mp_b700_s1", 700, 8, 1, 16, 3, 64, 2, seed=28, with_o2=False, clamp_stress=True)   # both multi-path
    if "head_mid" in which:  # 16 < S + S(S+1)/2 <= 64 with hidden_dim <= 64: emission rows spread over the waves
        make_head_case("s5_h32_l1", 3, 21, 5, 12, 2, 32, 1, seed=23, with_o2=False)
        make_head_case("s9_h64_l2", 2, 19, 9, 8, 4, 64, 2, seed=24, with_o2=False, clamp_stress=True)
    ou, lv = example_sdes()
    if "elbo" in which:
        make_elbo_case("ou", ou, [0.0, 1.0, 2.0, 3.0, 4.0, 5.0], [[2.0], [1.5], [0.8], [1.2], [0.9], [1.1]],
                       0.1, Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3), [], [0, 2], 6, 100, 0.05, 21)
        make_elbo_case("lv", lv, [0.0, 1.0, 2.0, 3.0, 4.0],
                       [[71.0, 79.0], [47.61225908, 447.20971405], [80.53119269, 50.26254069],
                        [23.10087379, 339.40432691], [158.05238324, 66.79611979]],
                       1.0, Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3), [0, 1], [0, 1, 2],
                       5, 40, 0.1, 22)
    if "encoder" in which:
        make_encoder_case()
    if "manifest" in which:
        make_manifest()
    if "trajectory" in which:
        make_trajectory()
    if "fused_dims" in which:
        make_fused_dims()
    if "encoder_d128" in which:
        make_encoder_d128()
    if "em" in which:
        make_em_cases()
    if "sde_coeffs" in which:
        make_sde_coeffs()
    if "posterior_sample" in which:
        make_posterior_sample()
    if "pretrain" in which:
        make_pretrain()
