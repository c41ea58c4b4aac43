"""Encoder parity at head_dim 128 and at the dims of BASELINE config 5.

* ``tests/golden/encoder_d128.npz`` (``make_golden.py encoder_d128``, the imported reference on CPU, fp32):
  ``ObservationContextEncoder`` (reference models/encoder.py:58-99, primitives/sit.py:78-186, attn.py:71-117) at hidden 128 /
  ONE head (head_dim 128) / depth 2, batch 104 x 41 tokens = 4264 rows: forward + gradients w.r.t. theta and every parameter.
  On the GPU these dims run the streamed attention kernels (D = 128), the 64-pair QK-norm / RoPE kernels, the 128-wide gate
  and the packed bf16 Linears with the own MFMA GEMMs.
* the module at config-5 dims (hidden 512 / 4 heads / depth 2, 1001 grid tokens, batch 5 = 5005 rows): K = 512 rows GEMMs with
  the SwiGLU epilogues, deep reductions, streamed attention at N = 1001 -- fused bf16 route against the fp32 torch chain of the
  same module (forward and ALL gradients) and, tightly, against the bf16 torch chain.

Tolerances (relative to the max magnitude of the compared tensor):
  CPU unfused chain vs reference, fp32 .............. context 1e-4, gradients 1e-3
  GPU fused fp32 vs reference ....................... context 2e-5, gradients 2e-4
  GPU fused bf16 vs reference (fp32) ................ context 3e-2, gradients 8e-2 (scalar parameters 0.3), as test_fused_dims
  GPU fused bf16 vs torch bf16 chain ................ relative L2 error (||a - b|| / ||b||): context 5e-3, gradients 1.5e-2.  A
      max-norm bound cannot go below one bf16 ulp of the largest element (2^-8 = 3.9e-3 .. 7.8e-3: two bf16 implementations
      differ from each other by as much as either differs from fp32), the L2 norm averages the rounding noise instead: a
      bf16-only kernel bug of a few % of the signal is far outside it
"""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err
from test_host_logic import _load_sd


def _fixture():
    return dict(np.load(f"{GOLDEN}/encoder_d128.npz"))


def _encoder(d, device):
    from viforsdes_amd import EncoderConfig
    from viforsdes_amd.models.encoder import ObservationContextEncoder
    hid, cond, heads, depth, _ = (int(v) for v in d["cfg"])
    enc = ObservationContextEncoder(2, 3, EncoderConfig(hidden_dim=hid, cond_dim=cond, num_heads=heads, depth=depth))
    enc.load_state_dict(_load_sd(d, "sd::"), strict=True)
    return enc.to(device).train()


def _run(enc, obs_v, obs_t, theta, horizon, dt, gout, autocast):
    dev = theta.device
    th = theta.detach().clone().requires_grad_(True)
    with torch.autocast(device_type=dev.type, dtype=torch.bfloat16, enabled=autocast):
        ctx = enc(obs_v.to(dev), obs_t.to(dev), th, horizon, dt)
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    params = [p for n, p in enc.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad((ctx.float() * gout).sum(), [th] + params)
    out = {"theta": grads[0].detach().float().cpu().numpy()}
    for n, g in zip(names, grads[1:]):
        out[n] = g.detach().float().cpu().numpy()
    return ctx.detach().float().cpu().numpy(), out


def _l2(a, b):
    return float(np.linalg.norm((np.asarray(a, np.float64) - b).ravel()) / (np.linalg.norm(np.asarray(b, np.float64).ravel()) + 1e-300))


def _compare(tag, ctx, grads, ctx_ref, grads_ref, tol_ctx, tol_grad, tol_scalar=None, metric=rel_err):
    tol_scalar = tol_grad if tol_scalar is None else tol_scalar
    e_ctx = metric(ctx, ctx_ref)
    errs = {n: metric(g, grads_ref[n]) for n, g in grads.items()}
    ranked = sorted(errs, key=errs.get, reverse=True)
    print(f"\n{tag}: context err {e_ctx:.2e}; gradient errors: median {float(np.median(list(errs.values()))):.2e}, largest "
          + ", ".join(f"{n} {errs[n]:.2e}" for n in ranked[:5]))
    assert e_ctx < tol_ctx, e_ctx
    for n in ranked:
        tol = tol_scalar if (n != "theta" and grads_ref[n].size == 1) else tol_grad
        assert errs[n] < tol, (n, errs[n], tol)


def _fixture_case(device, autocast, tol_ctx, tol_grad, tol_scalar=None):
    d = _fixture()
    enc = _encoder(d, device)
    B = int(d["cfg"][4])
    theta = torch.from_numpy(d["theta"]).to(device)
    n_tok = int(round(float(d["time_horizon"]) / float(d["time_step"]))) + 1
    gout = torch.from_numpy(np.random.RandomState(int(d["g_context_seed"])).randn(B, n_tok, int(d["cfg"][0])).astype(np.float32)).to(device)
    ctx, grads = _run(enc, torch.from_numpy(d["obs_values"]), torch.from_numpy(d["obs_times"]), theta, float(d["time_horizon"]),
                      float(d["time_step"]), gout, autocast)
    ref = {"theta": d["grad_theta"], **{n: d["grad::" + n] for n in grads if n != "theta"}}
    _compare(f"encoder d128 ({device}, autocast={autocast}) vs reference", ctx[d["context_rows"]], grads, d["context"], ref,
             tol_ctx, tol_grad, tol_scalar)


def test_encoder_d128_unfused_chain_matches_reference_cpu():
    _fixture_case("cpu", False, 1e-4, 1e-3)


@pytest.mark.gpu
def test_encoder_d128_fused_fp32_matches_reference_gpu():
    from viforsdes_amd.primitives import fused
    assert fused.usable(torch.empty(2, 41, 128, device="cuda:0"), 128, 128), "head_dim 128 must take the fused route"
    _fixture_case("cuda:0", False, 2e-5, 2e-4)


@pytest.mark.gpu
def test_encoder_d128_fused_bf16_matches_reference_gpu():
    from viforsdes_amd.primitives import fused
    assert fused.attention_usable(torch.empty(2, 41, 1, 128, device="cuda:0", dtype=torch.bfloat16))
    _fixture_case("cuda:0", True, 3e-2, 8e-2, tol_scalar=0.3)


def _fused_vs_torch(enc, obs_v, obs_t, theta, horizon, dt, gout, tag, tol_ctx32, tol_grad32, tol_ctx16, tol_grad16):
    """fused bf16 route vs (a) the fp32 torch chain and (b) the bf16 torch chain of the same module."""
    from viforsdes_amd.primitives import fused
    c_f, g_f = _run(enc, obs_v, obs_t, theta, horizon, dt, gout, True)
    fused.ENABLED = False
    try:
        c_32, g_32 = _run(enc, obs_v, obs_t, theta, horizon, dt, gout, False)
        c_16, g_16 = _run(enc, obs_v, obs_t, theta, horizon, dt, gout, True)
    finally:
        fused.ENABLED = True
    _compare(f"{tag}: fused bf16 vs fp32 torch chain", c_f, g_f, c_32, g_32, tol_ctx32, tol_grad32, tol_scalar=0.3)
    _compare(f"{tag}: fused bf16 vs bf16 torch chain (relative L2)", c_f, g_f, c_16, g_16, tol_ctx16, tol_grad16, tol_scalar=0.15,
             metric=_l2)


@pytest.mark.gpu
def test_fused_bf16_tracks_the_bf16_torch_chain_at_fixture_dims():
    """The two reference-anchored fixtures' dims (head_dim 64 x 2 heads, head_dim 128 x 1 head): tight bf16-vs-bf16 bound."""
    for name, prefix in (("fused_dims.npz", "init::encoder."), ("encoder_d128.npz", "sd::")):
        d = dict(np.load(f"{GOLDEN}/{name}"))
        from viforsdes_amd import EncoderConfig
        from viforsdes_amd.models.encoder import ObservationContextEncoder
        heads = 2 if name == "fused_dims.npz" else 1
        enc = ObservationContextEncoder(2, 3, EncoderConfig(hidden_dim=128, cond_dim=16, num_heads=heads, depth=2))
        enc.load_state_dict(_load_sd(d, prefix), strict=True)
        enc = enc.to("cuda:0").train()
        theta = torch.from_numpy(d["enc_theta"] if "enc_theta" in d else d["theta"]).to("cuda:0")
        gout = torch.from_numpy(np.random.RandomState(5).randn(theta.shape[0], 41, 128).astype(np.float32)).to("cuda:0")
        _fused_vs_torch(enc, torch.from_numpy(d["obs_values"]), torch.from_numpy(d["obs_times"]), theta, 2.0, 0.05, gout, name,
                        3e-2, 8e-2, 5e-3, 1.5e-2)


@pytest.mark.gpu
def test_encoder_module_at_config5_dims():
    """hidden 512 / 4 heads (head_dim 128) / depth 2, 1001 grid tokens (horizon 10, dt 0.01), 11 observations of an 8-dim state,
    16 SDE parameters, batch 5: the encoder of BASELINE config 5 with a shortened depth, forward and all gradients."""
    from viforsdes_amd import EncoderConfig
    from viforsdes_amd.models.encoder import ObservationContextEncoder
    from viforsdes_amd.primitives import fused
    dev = "cuda:0"
    torch.manual_seed(512)
    enc = ObservationContextEncoder(8, 16, EncoderConfig(hidden_dim=512, num_heads=4, depth=2)).to(dev).train()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():   # zero-initialised modulators / gates would make the blocks the identity
        for n, p in enc.named_parameters():
            if p.requires_grad and (float(p.abs().sum()) == 0.0 or "v_residual_lambda" in n):
                p.add_((torch.randn(p.shape, generator=g) * 0.1).to(dev))
    B, n_obs = 5, 11
    obs_t = torch.linspace(0.0, 10.0, n_obs)
    obs_v = torch.sin(torch.arange(n_obs * 8, dtype=torch.float32)).reshape(n_obs, 8)
    theta = torch.randn(B, 16, generator=g).to(dev)
    gout = torch.randn(B, 1001, 512, generator=g).to(dev)
    x = torch.empty(B, 1001, 512, device=dev, dtype=torch.bfloat16)
    assert fused.usable(x, 512, 128) and fused.attention_usable(torch.empty(B, 1001, 4, 128, device=dev, dtype=torch.bfloat16))
    assert fused.packed_linear_usable(x, 3 * 512 + 128, 512) and fused.swiglu_mlp_usable(x, 1408), \
        "config-5 dims must run the K = 512 rows GEMMs with the SwiGLU epilogues"
    _fused_vs_torch(enc, obs_v, obs_t, theta, 10.0, 0.01, gout, "config-5 dims (512 / 4 heads / N = 1001)", 3e-2, 8e-2, 5e-3, 1.5e-2)
