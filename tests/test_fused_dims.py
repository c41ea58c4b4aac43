"""Reference-anchored parity at dims where the fused encoder route is ACTIVE.

``tests/golden/fused_dims.npz`` (made by ``make_golden.py fused_dims`` from the imported reference, CPU fp32): encoder
hidden 128 / 2 heads (head_dim 64) / depth 2, GRU 64 x 2, Lotka-Volterra, 41 grid tokens, batch 104 -- i.e. C % 64 == 0,
head_dim 64 and 104 * 41 = 4264 >= 4096 token rows, so on the GPU every fused elementwise op, the own attention kernels
(bf16) and the packed bf16 Linears + weight-gradient kernel take part.  One initial state serves an encoder
forward/gradient case and a K=20-step trainer trajectory with the reference's recorded draws.

Tolerances (relative to the max magnitude of the compared tensor unless stated):
  CPU (unfused torch chains + C oracle), fp32 ............ encoder 1e-4 / grads 1e-3; ELBO per step 2e-4 rel
  GPU fused route, fp32 (mixed precision off) ............ encoder 2e-5 / grads 2e-4; ELBO per step 1e-5 rel,
                                                           final means / log-stds / expected_value / EMA 1e-4
                                                           (measured: 3e-7 / 6e-6; 2e-7; 2e-7)
  GPU fused route, bf16 autocast (the benchmark setting) . encoder 3e-2 / grads 8e-2 (bf16 activations: 8 mantissa bits
                                                           through 2 blocks; the scalar v_residual_lambda 0.3); ELBO per step
                                                           6e-2 rel of |ELBO| (the K=20 bf16 trajectory drifts from the fp32
                                                           one step by step), final expected_value 5e-2 rel
"""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err
from test_host_logic import _load_sd

K_STEPS = 20


def _fixture():
    return dict(np.load(f"{GOLDEN}/fused_dims.npz"))


def _trainer(d, device, mixed_precision):
    from viforsdes_amd import EncoderConfig, GaussianObservationLikelihood, HeadConfig, Observations, Prior, PriorType, TrainingConfig
    from viforsdes_amd.console import Console
    from viforsdes_amd.examples.sdes import LotkaVolterra
    from viforsdes_amd.inference.trainer import VariationalInferenceTrainer
    K, B = (int(v) for v in d["cfg"])
    obs = Observations(times=torch.from_numpy(d["obs_times"]), values=torch.from_numpy(d["obs_values"]))
    tr = VariationalInferenceTrainer(
        sde=LotkaVolterra(), observations=obs, observation_likelihood=GaussianObservationLikelihood(variance=0.25),
        prior=Prior(type=PriorType.LOG_NORMAL, mean=0.0, std=1.5, dim=3), time_horizon=float(d["horizon"]),
        config=TrainingConfig(time_step=float(d["dt"]), batch_size=B, n_iterations=K, learning_rate=1e-3, sde_param_lr=1e-2),
        encoder_config=EncoderConfig(hidden_dim=128, cond_dim=16, num_heads=2, depth=2),
        head_config=HeadConfig(hidden_dim=64, num_layers=2), state_positive_dims=[0, 1], sde_param_positive_dims=[0, 1, 2],
        device=device, mixed_precision=mixed_precision, console=Console(enabled=False))
    tr.ctx.model.load_state_dict(_load_sd(d, "init::"))
    tr.ctx.ema._init_shadow()
    tr.ctx.model.train()
    return tr, obs


def _encoder_case(d, device, autocast, tol_ctx, tol_grad, tol_scalar=None):
    tol_scalar = tol_grad if tol_scalar is None else tol_scalar
    tr, obs = _trainer(d, device, False)
    enc = tr.ctx.model.encoder
    dev = torch.device(device)
    theta = torch.from_numpy(d["enc_theta"]).to(dev).requires_grad_(True)
    with torch.autocast(device_type=dev.type, dtype=torch.bfloat16, enabled=autocast):
        ctx = enc(obs.values.to(dev), obs.times.to(dev), theta, float(d["horizon"]), float(d["dt"]))
    rows = d["enc_context_rows"]
    e_ctx = rel_err(ctx.detach().float().cpu().numpy()[rows], d["enc_context"])
    gout = torch.from_numpy(np.random.RandomState(int(d["enc_g_context_seed"])).randn(*ctx.shape).astype(np.float32)).to(dev)
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    params = [p for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx.float() * gout).sum(), [theta] + params)
    errs = {"theta": rel_err(grads[0].float().cpu().numpy(), d["enc_grad_theta"])}
    for n, g in zip(names, grads[1:]):
        errs[n] = rel_err(g.float().cpu().numpy(), d["enc_grad::" + n])
    ranked = sorted(errs, key=errs.get, reverse=True)
    print(f"\nencoder ({device}, autocast={autocast}): context err {e_ctx:.2e}; gradient errors: median "
          f"{float(np.median(list(errs.values()))):.2e}, largest " + ", ".join(f"{n} {errs[n]:.2e}" for n in ranked[:6]))
    assert e_ctx < tol_ctx, e_ctx
    for n in ranked:
        # a scalar parameter's gradient (v_residual_lambda) is ONE sum with cancellation over all tokens: in bf16 its relative
        # error is that of the upstream bf16 gradients times the cancellation factor, so it gets its own (stated) bound
        tol = tol_scalar if (n != "theta" and d["enc_grad::" + n].size == 1) else tol_grad
        assert errs[n] < tol, (n, errs[n], tol)


def _trajectory(d, device, mixed_precision, tol_elbo, tol_final, steps=K_STEPS):
    tr, _ = _trainer(d, device, mixed_precision)
    dev = torch.device(device)
    worst, per_step = 0.0, []
    for k in range(steps):
        r = tr._train_step(tr.ctx.model, theta_eps=torch.from_numpy(d["theta_eps"][k]).to(dev),
                           path_noise=torch.from_numpy(d["path_noise"][k]).to(dev))
        tr.ctx.ema.update()
        e = abs(float(r.elbo_result.evidence_lower_bound) - d["elbo"][k]) / abs(d["elbo"][k])
        worst = max(worst, e)
        per_step.append(e)
    post = tr.ctx.model.sde_parameter_posterior
    out = {"elbo": worst}
    if steps == int(d["cfg"][0]):
        out["mean"] = rel_err(post.mean.detach().cpu().numpy(), d["final_mean"])
        out["log_std"] = rel_err(post.log_std.detach().cpu().numpy(), d["final_log_std"])
        out["expected_value"] = rel_err(post.expected_value.detach().cpu().numpy(), d["final_expected_value"])
        out["ema_mean"] = rel_err(tr.ctx.ema.shadow["sde_parameter_posterior.mean"].cpu().numpy(), d["ema_mean"])
    print(f"\ntrajectory ({device}, mixed_precision={mixed_precision}): " + " ".join(f"{k_}={v:.2e}" for k_, v in out.items())
          + " per-step ELBO err " + " ".join(f"{e:.1e}" for e in per_step))
    for k_, v in out.items():
        assert v < (tol_elbo if k_ == "elbo" else tol_final), (k_, v)
    return tr


# ------------------------------------------------------------------------------------------- CPU (host logic + oracle)
@pytest.fixture()
def oracle_backend():
    from oracle.torch_backend import OracleBackend
    from viforsdes_amd.kernels.backend import set_backend
    set_backend(OracleBackend())
    yield
    set_backend(None)


def test_encoder_unfused_chain_matches_reference_cpu():
    _encoder_case(_fixture(), "cpu", False, 1e-4, 1e-3)


def test_trajectory_first_steps_cpu(oracle_backend):
    _trajectory(_fixture(), "cpu", False, 2e-4, 1e-3, steps=4)   # 4 of the 20 steps: keeps the CPU suite short


# ----------------------------------------------------------------------------------------------------- GPU (fused route)
def _assert_fused_route_active(tr, bf16):
    from viforsdes_amd.primitives import fused
    enc = tr.ctx.model.encoder
    C, hd = enc.hidden_dim, enc.hidden_dim // enc.num_heads
    x = torch.empty(2, 41, C, device="cuda:0", dtype=torch.bfloat16 if bf16 else torch.float32)
    assert fused.ENABLED and fused.usable(x, C, hd), "the fixture's dims must take the fused encoder route"
    if bf16:
        assert fused.attention_usable(torch.empty(2, 41, enc.num_heads, hd, device="cuda:0", dtype=torch.bfloat16))
        assert fused.packed_linear_usable(torch.empty(104, 41, C, device="cuda:0", dtype=torch.bfloat16), 3 * C + hd, C)
        # ... and, in the training step, the attention block's fused core (projection epilogue + attention-kernel epilogues):
        # the reference fixture below is what pins its forward and backward
        att = enc.sit.blocks[1].self_attn
        xb = torch.zeros(104, 41, C, device="cuda:0", dtype=torch.bfloat16)
        from viforsdes_amd.primitives.embeddings import RotarySpec
        rot = RotarySpec.from_freqs(enc.rope_freqs[:41].to("cuda:0"))
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            att.forward_fused(xb, rotary=rot, v0=None)   # builds the packed [q | k | v | gate] operand
        assert fused.attention_core_usable(xb, att._proj_pack, enc.num_heads, hd, att.q_norm.weight, att.k_norm.weight,
                                           rot.cos_sin_tables(41)[0]), "the fixture's dims must take the fused attention core"


@pytest.mark.gpu
def test_encoder_fused_fp32_matches_reference_gpu():
    d = _fixture()
    _assert_fused_route_active(_trainer(d, "cuda:0", False)[0], False)
    _encoder_case(d, "cuda:0", False, 2e-5, 2e-4)


@pytest.mark.gpu
def test_encoder_fused_bf16_matches_reference_gpu():
    d = _fixture()
    _assert_fused_route_active(_trainer(d, "cuda:0", True)[0], True)
    _encoder_case(d, "cuda:0", True, 3e-2, 8e-2, tol_scalar=0.3)


@pytest.mark.gpu
def test_trajectory_fused_fp32_gpu():
    _trajectory(_fixture(), "cuda:0", False, 1e-5, 1e-4)


@pytest.mark.gpu
def test_trajectory_fused_bf16_autocast_gpu():
    _trajectory(_fixture(), "cuda:0", True, 6e-2, 5e-2)
