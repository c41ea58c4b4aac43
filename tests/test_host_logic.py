"""CPU: host-side logic of the drop-in (configs, validation, encoder, checkpoint layout, trainer).

Where an operator of the fused head / ELBO is needed the CPU oracle is plugged in through
viforsdes_amd.kernels.backend.set_backend (test infrastructure; the shipped default is HIP)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err
from viforsdes_amd import (EncoderConfig, GaussianObservationLikelihood, HeadConfig, InferenceConfig, Observations,
                           PretrainConfig, Prior, PriorType, TrainingConfig, make_sde)
from viforsdes_amd.console import Console
from viforsdes_amd.examples.sdes import LotkaVolterra, OrnsteinUhlenbeck
from viforsdes_amd.infer import validate_inference_inputs
from viforsdes_amd.inference.state_space import StateSpace
from viforsdes_amd.inference.trainer import VariationalInferenceTrainer
from viforsdes_amd.kernels.backend import set_backend
from viforsdes_amd.models.encoder import ObservationContextEncoder
from viforsdes_amd.models.sde_parameter_posterior import SDEParameterPosterior
from viforsdes_amd.models.variational_sde_posterior import VariationalSDEPosterior
from viforsdes_amd.posterior.variational_posterior import VariationalPosterior


@pytest.fixture()
def oracle_backend():
    from oracle.torch_backend import OracleBackend
    set_backend(OracleBackend())
    yield
    set_backend(None)


def _load_sd(d, prefix):
    sd = {}
    for k, v in d.items():
        if k.startswith(prefix):
            t = torch.from_numpy(v)
            if k.endswith("rope_freqs"):
                t = torch.view_as_complex(t.contiguous())
            sd[k[len(prefix):]] = t
    return sd


def test_config_defaults_and_validation(tmp_path):
    t = TrainingConfig()
    assert (t.time_step, t.batch_size, t.n_iterations, t.learning_rate, t.sde_param_lr, t.grad_clip_norm) == \
        (0.1, 50, 25000, 1e-4, 1e-3, 1.0)
    assert t.amp_dtype.value == torch.bfloat16
    e = EncoderConfig()
    assert (e.hidden_dim, e.cond_dim, e.num_heads, e.depth) == (128, 128, 4, 4) and abs(e.mlp_ratio - 8 / 3) < 1e-12
    assert (HeadConfig().hidden_dim, HeadConfig().num_layers) == (64, 2)
    p = PretrainConfig()
    assert (p.n_iterations, p.batch_size, p.learning_rate, p.init_scale) == (1000, 4096, 0.02, 2.0)
    for bad in (dict(time_step=0.0), dict(batch_size=0), dict(learning_rate=-1.0)):
        with pytest.raises(ValueError):
            TrainingConfig(**bad)
    with pytest.raises(ValueError):
        EncoderConfig(hidden_dim=130, num_heads=4)
    with pytest.raises(ValueError):
        HeadConfig(num_layers=0)
    y = tmp_path / "c.yaml"
    y.write_text("batch_size: 7\ntime_step: 0.25\n")
    c = TrainingConfig.from_yaml(y)
    assert c.batch_size == 7 and c.time_step == 0.25
    with pytest.raises(Exception):
        c.batch_size = 3  # frozen


def test_problem_types_validate():
    with pytest.raises(ValueError):
        Observations(times=torch.tensor([0.0, 2.0, 1.0]), values=torch.zeros(3, 1))
    with pytest.raises(ValueError):
        Observations(times=torch.zeros(2, 1), values=torch.zeros(2, 1))
    with pytest.raises(ValueError):
        Observations(times=torch.zeros(2), values=torch.zeros(3, 1))
    with pytest.raises(ValueError):
        GaussianObservationLikelihood(variance=0.0)
    with pytest.raises(ValueError):
        Prior(type=PriorType.NORMAL, mean=0.0, std=-1.0, dim=2)
    with pytest.raises(ValueError):
        StateSpace(2, [0, 0])
    lik = GaussianObservationLikelihood(variance=0.5, obs_matrix=torch.tensor([[1.0, 0.0]]))
    lp = lik.log_prob(torch.tensor([[0.3]]), torch.tensor([[0.1, 5.0]]))
    assert abs(float(lp) - (-0.5 * 0.04 / 0.5 - 0.5 * np.log(2 * np.pi * 0.5))) < 1e-6
    sde = make_sde(lambda x, th: -x, lambda x, th: torch.eye(1).expand(x.shape[0], 1, 1), 1, 1)
    assert sde.drift(torch.ones(2, 1), torch.ones(2, 1)).shape == (2, 1)


def test_infer_input_rules():
    obs = Observations(times=torch.tensor([0.0, 1.0]), values=torch.zeros(2, 1))
    prior = Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=3)
    ok = dict(observations=obs, time_horizon=1.0, time_step=0.1, state_dim=1, sde_param_dim=3, state_positive_dims=[],
              sde_param_positive_dims=[0], prior=prior)
    validate_inference_inputs(**ok)
    cases = [dict(time_horizon=1.05), dict(time_step=0.3), dict(state_positive_dims=[1]), dict(sde_param_positive_dims=[0, 0]),
             dict(prior=Prior(type=PriorType.NORMAL, mean=0.0, std=1.0, dim=2)),
             dict(observations=Observations(times=torch.tensor([0.5, 1.0]), values=torch.zeros(2, 1))),
             dict(observations=Observations(times=torch.tensor([0.0, 0.55]), values=torch.zeros(2, 1))),
             dict(observations=Observations(times=torch.tensor([0.0, 2.0]), values=torch.zeros(2, 1)))]
    for change in cases:
        with pytest.raises(ValueError):
            validate_inference_inputs(**{**ok, **change})
    assert InferenceConfig().device == "cuda" and InferenceConfig().mixed_precision is True


@pytest.mark.parametrize("name", ["ou", "lv"])
def test_theta_densities_and_state_space_vs_golden(name):
    d = dict(np.load(f"{GOLDEN}/elbo_{name}.npz"))
    P = int(d["dims"][3])
    post = SDEParameterPosterior(P, [int(v) for v in d["theta_positive_dims"]])
    with torch.no_grad():
        post.mean.copy_(torch.from_numpy(d["q_mean"])); post.log_std.copy_(torch.from_numpy(d["q_log_std"]))
    theta = post.rsample(int(d["dims"][0]), eps=torch.from_numpy(d["eps_theta"]))
    assert rel_err(theta.detach().numpy(), d["theta"]) < 1e-6
    assert rel_err(post.log_prob(theta).detach().numpy(), d["post_lp"]) < 1e-6
    assert rel_err(post.expected_value.detach().numpy(), d["expected_value"]) < 1e-6
    prior = Prior(type=PriorType.LOG_NORMAL if int(d["prior_type"]) else PriorType.NORMAL, mean=float(d["prior_mean"]),
                  std=float(d["prior_std"]), dim=P)
    assert rel_err(prior.log_prob(theta).detach().numpy(), d["prior_lp"]) < 1e-6
    space = StateSpace(int(d["dims"][2]), [int(v) for v in d["state_positive_dims"]])
    z = torch.from_numpy(d["z"])
    assert rel_err(space.to_state(z).numpy(), d["x"]) < 1e-6
    assert rel_err(space.to_latent(torch.from_numpy(d["x0"])).numpy(), d["z0"]) < 1e-6
    if d["jac"].any():
        assert rel_err(space.log_jacobian(z[:, 1:]).sum(-1).numpy(), d["jac"]) < 1e-6
    sde = LotkaVolterra() if name == "lv" else OrnsteinUhlenbeck()
    B, T, S, _ = (int(v) for v in d["dims"])
    xt = torch.from_numpy(d["x"])[:, :-1].reshape(B * T, S)
    th = torch.from_numpy(d["theta"]).unsqueeze(1).expand(B, T, P).reshape(B * T, P)
    assert rel_err(sde.drift(xt, th).reshape(B, T, S).numpy(), d["drift"]) < 1e-6
    assert rel_err(sde.diffusion(xt, th).reshape(B, T, S, S).numpy(), d["diffusion"]) < 1e-6


def test_encoder_matches_reference_forward_and_gradients():
    d = dict(np.load(f"{GOLDEN}/encoder_tiny.npz"))
    hid, cond, heads, depth = (int(v) for v in d["cfg"])
    enc = ObservationContextEncoder(2, 3, EncoderConfig(hidden_dim=hid, cond_dim=cond, num_heads=heads, depth=depth))
    enc.load_state_dict(_load_sd(d, "sd::"), strict=True)
    theta = torch.from_numpy(d["theta"]).requires_grad_(True)
    ctx = enc(torch.from_numpy(d["obs_values"]), torch.from_numpy(d["obs_times"]), theta, float(d["time_horizon"]),
              float(d["time_step"]))
    assert rel_err(ctx.detach().numpy(), d["context"]) < 2e-6
    named = [(n, p) for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx * torch.from_numpy(d["g_context"])).sum(), [theta] + [p for _, p in named])
    assert rel_err(grads[0].numpy(), d["grad_theta"]) < 2e-5
    for (n, _), g in zip(named, grads[1:]):
        assert rel_err(g.numpy(), d["grad::" + n]) < 5e-5, n


def test_state_dict_layout_matches_reference_manifest():
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))
    for depth in (2, 8):
        m = VariationalSDEPosterior(2, 2, 3, EncoderConfig(hidden_dim=256, num_heads=4, depth=depth),
                                    HeadConfig(hidden_dim=64, num_layers=2), [0, 1, 2])
        mine = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
        assert mine == man[f"lv_depth{depth}"]
        assert [n for n, _ in m.named_parameters()] == man[f"lv_depth{depth}_params"]
    m = VariationalSDEPosterior(1, 1, 3, EncoderConfig(), HeadConfig(hidden_dim=32, num_layers=3), [0, 2])
    assert {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()} == man["ou_default_l3"]
    # checkpoints written with torch.compile on (encoder.sit._orig_mod.*) load too
    sd = {k.replace("encoder.sit.", "encoder.sit._orig_mod."): v for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    b = m.head.out_proj.bias
    assert float(b[0]) == 0.0 and float(b[1]) == 1.0 and float(m.head.out_proj.weight.abs().sum()) == 0.0


def _tiny_trainer(d):
    K, B = (int(v) for v in d["cfg"])
    obs = Observations(times=torch.from_numpy(d["obs_times"]), values=torch.from_numpy(d["obs_values"]))
    tr = VariationalInferenceTrainer(
        sde=LotkaVolterra(), observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=0.25),
        prior=Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3), time_horizon=float(d["horizon"]),
        config=TrainingConfig(time_step=float(d["dt"]), batch_size=B, n_iterations=K, learning_rate=1e-3, sde_param_lr=1e-2),
        encoder_config=EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=2),
        head_config=HeadConfig(hidden_dim=16, num_layers=2), state_positive_dims=[0, 1], sde_param_positive_dims=[0, 1, 2],
        device="cpu", mixed_precision=False, console=Console(enabled=False))
    tr.ctx.model.load_state_dict(_load_sd(d, "init::"))
    tr.ctx.ema._init_shadow()
    tr.ctx.model.train()
    return tr, K, obs


def test_training_trajectory_matches_reference(oracle_backend):
    """K optimizer steps with the reference's recorded theta/path noise: ELBO, components, gradient
    norm per step, final posterior parameters, expected_value and EMA (fp32 tolerance 2e-5 rel)."""
    d = dict(np.load(f"{GOLDEN}/trajectory_tiny.npz"))
    tr, K, _ = _tiny_trainer(d)
    model = tr.ctx.model
    for k in range(K):
        r = tr._train_step(model, theta_eps=torch.from_numpy(d["theta_eps"][k]), path_noise=torch.from_numpy(d["path_noise"][k]))
        tr.ctx.ema.update()
        c = r.elbo_result.components
        comps = [float(v) for v in (c.observation_log_prob, c.sde_log_prob, c.generative_log_prob, c.prior_log_prob,
                                    c.posterior_log_prob)]
        assert abs(float(r.elbo_result.evidence_lower_bound) - d["elbo"][k]) < 2e-5 * abs(d["elbo"][k]), k
        assert np.allclose(comps, d["components"][k], rtol=2e-5, atol=1e-5), k
        assert abs(float(r.grad_norm) - d["grad_norm"][k]) < 1e-4 * d["grad_norm"][k], k
    post = model.sde_parameter_posterior
    assert np.allclose(post.mean.detach().numpy(), d["final_mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(post.log_std.detach().numpy(), d["final_log_std"], rtol=1e-4, atol=1e-6)
    assert np.allclose(post.expected_value.detach().numpy(), d["final_expected_value"], rtol=1e-4)
    assert np.allclose(tr.ctx.ema.shadow["sde_parameter_posterior.mean"].numpy(), d["ema_mean"], rtol=1e-4, atol=1e-7)
    assert np.allclose(model.head.out_proj.bias.detach().numpy(), d["final::head.out_proj.bias"], rtol=1e-4, atol=1e-6)
    assert np.allclose(model.encoder.bridge_token.detach().numpy(), d["final::encoder.bridge_token"], rtol=1e-4, atol=1e-6)


def test_train_loop_history_callback_and_checkpoint_roundtrip(oracle_backend, tmp_path):
    d = dict(np.load(f"{GOLDEN}/trajectory_tiny.npz"))
    tr, K, obs = _tiny_trainer(d)
    seen = []
    torch.manual_seed(0)
    state = tr.train(callback=lambda step, elbo: seen.append((step, elbo)))
    assert len(state.evidence_lower_bound_history) == K and [s for s, _ in seen] == list(range(K))
    assert state.best_evidence_lower_bound == max(state.evidence_lower_bound_history)
    prior = Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3)
    vp = VariationalPosterior(model=state.model, exponential_moving_average=state.exponential_moving_average, prior=prior,
                              observations=obs, time_horizon=float(d["horizon"]), time_step=float(d["dt"]),
                              state_space=StateSpace(2, [0, 1]), evidence_lower_bound_history=state.evidence_lower_bound_history,
                              device=torch.device("cpu"))
    before = {k: v.clone() for k, v in state.model.state_dict().items()}
    s = vp.sample(5)
    assert s.sde_parameters.shape == (5, 3) and s.diffusion_paths.shape == (5, 11, 2) and (s.diffusion_paths > 0).all()
    for k, v in state.model.state_dict().items():  # EMA swap restored the live weights
        assert torch.equal(v, before[k])
    summ = vp.summary(16)
    assert summ.sde_parameter_quantiles.q50.shape == (3,) and summ.diffusion_path_mean.shape == (11, 2)
    assert vp.diagnostics().n_iterations == K
    path = tmp_path / "post.pt"
    vp.save(path)
    raw = torch.load(path, weights_only=True)
    assert sorted(raw) == ["ema_state", "evidence_lower_bound_history", "model_state", "state_positive_dims", "time_horizon",
                           "time_step"]
    assert set(raw["ema_state"]) == {n for n, _ in state.model.named_parameters()}
    fresh = VariationalSDEPosterior(2, 2, 3, EncoderConfig(hidden_dim=32, cond_dim=16, num_heads=4, depth=2),
                                    HeadConfig(hidden_dim=16, num_layers=2), [0, 1, 2])
    vp2 = VariationalPosterior.load(path, fresh, prior, obs, torch.device("cpu"))
    assert vp2.time_step == float(d["dt"]) and vp2.state_space.positive_dims == [0, 1]
    for k, v in vp2.model.state_dict().items():
        assert torch.equal(v, before[k])
    assert torch.equal(vp2.exponential_moving_average.shadow["head.out_proj.bias"],
                       state.exponential_moving_average.shadow["head.out_proj.bias"])


def test_pretraining_runs_and_returns_a_mean(oracle_backend):
    d = dict(np.load(f"{GOLDEN}/trajectory_tiny.npz"))
    tr, _, _ = _tiny_trainer(d)
    torch.manual_seed(1)
    mu = tr.pretrain_sde_parameters(PretrainConfig(n_iterations=5, batch_size=32))
    assert mu.shape == (3,) and torch.isfinite(mu).all()


def test_only_optimizers_that_own_packed_parameters_mark_packs_stale():
    """the process-wide ``Optimizer.step`` post-hook used to advance the parameter epoch for EVERY optimizer;
    a step of an unrelated model then cost every live pack a refresh."""
    import torch
    from viforsdes_amd.primitives import fused
    w = torch.nn.Parameter(torch.randn(8, 8))
    other = torch.nn.Parameter(torch.randn(4))
    pack = fused.PackedWeight(8, 8, [(w, 0, 8, 0)], None, "cpu")
    pack.operands()
    assert not pack.stale()
    o_other = torch.optim.SGD([other], lr=0.1)
    other.grad = torch.ones(4)
    o_other.step()
    assert not pack.stale()
    o_own = torch.optim.SGD([w], lr=0.1)
    w.grad = torch.ones(8, 8)
    o_own.step()
    assert pack.stale()


def test_latent_start_cache_follows_the_tensor():
    """sample_diffusion_paths remembers to_latent(x0) while x0 is the same unmodified memory (the start state is a constant of the
    problem); an in-place edit, another tensor, or a tensor that carries a gradient must not see a stale value."""
    import torch
    from viforsdes_amd.inference.diffusion_path_sampler import _latent_start
    from viforsdes_amd.inference.state_space import StateSpace
    ss = StateSpace(2, [0, 1])
    x0 = torch.tensor([[1.0, 2.0], [0.5, 3.0]])
    a = _latent_start(ss, x0)
    assert torch.equal(a, ss.to_latent(x0)) and _latent_start(ss, x0) is a                 # second call: the remembered tensor
    assert _latent_start(ss, x0[:1].expand(4, -1)).shape == (4, 2)                          # another layout of the same memory
    x0.mul_(2.0)                                                                            # in-place edit: version counter moves
    b = _latent_start(ss, x0)
    assert b is not a and torch.equal(b, ss.to_latent(x0))
    y = torch.tensor([[1.0, 2.0], [0.5, 3.0]], requires_grad=True)
    c = _latent_start(ss, y)
    assert c.requires_grad and torch.equal(c.detach(), ss.to_latent(y.detach()))            # gradients flow: never cached
    for k in range(8):                                                                      # fresh tensors: each gets its own value
        z = torch.full((2, 2), 1.0 + k)
        assert torch.equal(_latent_start(ss, z), ss.to_latent(z))


def test_three_bf16_pieces_reproduce_an_fp32_value_and_six_products_its_product():
    """The operand split of the head's weight-gradient tiles (vsde_tn_wide.hip::tw_split2), restated in numpy: x = h + m + l exactly
    with every piece a bf16 (round to nearest even of what is left), and hh + hm + mh + mm + hl + lh within 2^-23 of x y."""
    rng = np.random.default_rng(11)
    # exact while the smallest piece stays a normal number and the largest does not round up to infinity
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 20000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.38e38, 2.0 ** -109, 65504.0, 1 + 2.0 ** -23, 1 - 2.0 ** -24, 1 + 2.0 ** -8], np.float32)])

    def bf16(v):   # round to nearest even, result as fp32
        u = v.view(np.uint32).astype(np.uint64)
        u = (u + np.uint64(0x7fff) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xffff0000)
        return u.astype(np.uint32).view(np.float32)

    def split(v):
        hi = bf16(v)
        r = v - hi
        mid = bf16(r)
        s_ = r - mid
        lo = bf16(s_)
        return hi, mid, lo

    h, m, l = split(x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()
    y = rng.standard_normal(x.size).astype(np.float32)
    yh, ym, yl = split(y)
    f = lambda a: a.astype(np.float64)
    six = f(h) * f(yh) + f(h) * f(ym) + f(m) * f(yh) + f(m) * f(ym) + f(h) * f(yl) + f(l) * f(yh)
    exact = f(x) * f(y)
    ok = np.abs(exact) > 1e-300
    assert np.max(np.abs(six[ok] - exact[ok]) / np.abs(exact[ok])) < 2.0 ** -23
