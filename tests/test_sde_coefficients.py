"""Drift / diffusion of the built-in SDEs on all grid points (csrc/vsde_sde.hip: vsde_sde_coefficients_fwd/_bwd).

CPU: the float64 oracle against the golden vectors generated with the reference's example SDE classes and torch autograd
(tests/golden/make_golden.py::make_sde_coeffs), and against the package's own SDE classes (incl. the linear-diagonal benchmark
SDE, which the reference does not define).  GPU: the HIP kernels against the oracle on the golden inputs and at the LV
training size, and the ELBO with built-in coefficients against the ELBO through the Python callables."""
import os

import numpy as np
import pytest
import torch

from oracle import vsde_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sde_coefficients.npz")


def _gold(name):
    z = np.load(GOLD)
    return {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "_")}


@pytest.mark.parametrize("name", ["ou", "lv"])
def test_oracle_matches_reference_vectors(name):
    g = _gold(name)
    f, G = orc.sde_coefficients(name, g["x"], g["theta"])
    np.testing.assert_allclose(f, g["drift"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(G, g["diffusion"], rtol=2e-6, atol=1e-7)
    gx, gth = orc.sde_coefficients_bwd(name, g["x"], g["theta"], g["g_drift"], g["g_diffusion"])
    # fp32 autograd through 1 / sqrt(1e-6) factors: relative to the largest entry of each tensor
    assert np.abs(gx - g["grad_x"]).max() <= 2e-5 * np.abs(g["grad_x"]).max()
    assert np.abs(gth - g["grad_theta"]).max() <= 2e-5 * np.abs(g["grad_theta"]).max()


def _torch_reference(sde, x, theta, gf, gG):
    x = torch.tensor(x, dtype=torch.float64, requires_grad=True); theta = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
    B, T, S = x.shape[0], x.shape[1] - 1, x.shape[2]
    xf = x[:, :-1].reshape(B * T, S); tf = theta.unsqueeze(1).expand(B, T, -1).reshape(B * T, -1)
    f = sde.drift(xf, tf).reshape(B, T, S); G = sde.diffusion(xf, tf).reshape(B, T, S, S)
    gx, gth = torch.autograd.grad((f * torch.tensor(gf)).sum() + (G * torch.tensor(gG)).sum(), [x, theta])
    return f.detach().numpy(), G.detach().numpy(), gx.numpy(), gth.numpy()


@pytest.mark.parametrize("kind", ["ou", "lv", "linear_diagonal"])
def test_oracle_matches_package_sdes_f64(kind):
    from viforsdes_amd.examples.sdes import LinearDiagonalSDE, LotkaVolterra, OrnsteinUhlenbeck
    sde = {"ou": OrnsteinUhlenbeck(), "lv": LotkaVolterra(), "linear_diagonal": LinearDiagonalSDE(5)}[kind]
    rng = np.random.default_rng(7)
    B, T, S, P = 3, 6, sde.state_dim, sde.sde_param_dim
    x = rng.uniform(0.1, 3.0, (B, T + 1, S)); theta = rng.uniform(0.1, 0.9, (B, P))
    if kind == "linear_diagonal":
        theta[:, S:] = rng.normal(size=(B, S)) * 3.0
    gf = rng.normal(size=(B, T, S)); gG = rng.normal(size=(B, T, S, S))
    f, G, gx, gth = _torch_reference(sde, x, theta, gf, gG)
    of, oG = orc.sde_coefficients(kind, x, theta)
    ogx, ogth = orc.sde_coefficients_bwd(kind, x, theta, gf, gG)
    for a, b in ((of, f), (oG, G), (ogx, gx), (ogth, gth)):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-12)


_KIND = {"ou": "ornstein_uhlenbeck", "lv": "lotka_volterra", "linear_diagonal": "linear_diagonal"}


def _gpu_vs_oracle(kind, x, theta, gf, gG, tol):
    from viforsdes_amd import _hip
    dev = "cuda:0"
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    f, G = _hip.sde_coefficients_fwd(_KIND[kind], t(x), t(theta))
    gx, gth = _hip.sde_coefficients_bwd(_KIND[kind], t(x), t(theta), t(gf), t(gG))
    x32, th32 = np.asarray(x, np.float32), np.asarray(theta, np.float32)
    of, oG = orc.sde_coefficients(kind, x32, th32)
    ogx, ogth = orc.sde_coefficients_bwd(kind, x32, th32, np.asarray(gf, np.float32), np.asarray(gG, np.float32))
    for got, want in ((f, of), (G, oG), (gx, ogx), (gth, ogth)):
        got = got.double().cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1.0), (kind, np.abs(got - want).max(), np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ou", "lv"])
def test_hip_matches_oracle_on_reference_inputs(name):
    g = _gold(name)
    # the clamp-boundary rows (1 / sqrt(1e-6) amplification) are part of these inputs
    _gpu_vs_oracle(name, g["x"], g["theta"], g["g_drift"], g["g_diffusion"], 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,B,T,S", [("lv", 512, 400, 2), ("ou", 128, 100, 1), ("linear_diagonal", 64, 1000, 8),
                                        ("linear_diagonal", 3, 7, 5), ("lv", 1, 1, 2)])
def test_hip_matches_oracle_at_size(kind, B, T, S):
    rng = np.random.default_rng(B + T)
    P = 2 * S if kind == "linear_diagonal" else 3
    x = rng.uniform(0.05, 200.0 if kind == "lv" else 3.0, (B, T + 1, S)); theta = rng.uniform(0.05, 0.9, (B, P))
    if kind == "linear_diagonal":
        theta[:, S:] = rng.normal(size=(B, S)) * 4.0
        theta[0, S] = 25.0   # softplus threshold branch
    gf = rng.normal(size=(B, T, S)); gG = rng.normal(size=(B, T, S, S))
    # g_theta sums T terms in fp32
    _gpu_vs_oracle(kind, x, theta, gf, gG, 5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["ou", "lv"])
def test_elbo_with_builtin_coefficients_matches_python_callables(problem):
    """Same ELBO value and gradients whether drift / diffusion come from the HIP kernels or the SDE's Python callables."""
    from viforsdes_amd.examples import sdes
    from viforsdes_amd.inference import evidence_lower_bound as elbo_mod
    sde = (sdes.ou_problem if problem == "ou" else sdes.lv_problem)()[0]
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    B, T, S, P = 16, 50, sde.state_dim, sde.sde_param_dim
    out = {}
    for flag in (True, False):
        elbo_mod.HIP_COEFFICIENTS = flag
        try:
            x = (torch.rand(B, T + 1, S, generator=torch.Generator().manual_seed(5)) * 2 + 0.3).to(dev).requires_grad_(True)
            theta = (torch.rand(B, P, generator=torch.Generator().manual_seed(6)) * 0.8 + 0.1).to(dev).requires_grad_(True)
            f, G = elbo_mod.sde_coefficients(sde, x, theta)
            w1 = torch.randn(f.shape, generator=torch.Generator().manual_seed(8)).to(dev)
            w2 = torch.randn(G.shape, generator=torch.Generator().manual_seed(9)).to(dev)
            loss = (f * w1).sum() + (G * w2).sum()
            gx, gth = torch.autograd.grad(loss, [x, theta])
            out[flag] = [t.detach().double().cpu() for t in (f, G, gx, gth)]
        finally:
            elbo_mod.HIP_COEFFICIENTS = True
    for a, b in zip(out[True], out[False]):
        assert (a - b).abs().max() <= 1e-5 * max(float(b.abs().max()), 1.0)
