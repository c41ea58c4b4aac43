"""Euler-Maruyama simulator of the model SDE (reference core/euler_maruyama.py:11-45) and its gradient.

CPU: the oracle's C restatement and the package's torch loop against the reference fixture (tests/golden/euler_maruyama.npz,
made by ``make_golden.py em``: example OU / LV SDEs, injected noise, LV rows that hit the 1e-6 clamp).
GPU: the HIP simulator (forward + reverse-mode backward through the C ABI) against the same fixture, against the float64
oracle at the pre-training size (B=4096, LV T=400), and the linear-diagonal kind against the torch loop.
Tolerances (relative to the max magnitude): forward 2e-6 vs the fp32 reference, gradients 2e-5; vs float64 at T=400: 2e-4 / 2e-4."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, rel_err

CASES = {"ou": "ornstein_uhlenbeck", "lv": "lotka_volterra"}


def _fx():
    return dict(np.load(f"{GOLDEN}/euler_maruyama.npz"))


def _sde(name):
    from viforsdes_amd.examples.sdes import LotkaVolterra, OrnsteinUhlenbeck
    return OrnsteinUhlenbeck() if name == "ou" else LotkaVolterra()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference(name):
    from oracle import vsde_oracle as vo
    d = _fx()
    pos, dt = list(d[f"{name}_pos"]), float(d[f"{name}_cfg"][1])
    tr = vo.euler_maruyama(name, d[f"{name}_x0"], d[f"{name}_theta"], d[f"{name}_noise"], dt, pos)
    gx, gt = vo.euler_maruyama_bwd(name, d[f"{name}_theta"], d[f"{name}_noise"], tr, d[f"{name}_g_traj"], dt, pos)
    assert rel_err(tr, d[f"{name}_traj"]) < 1e-6
    assert rel_err(gx, d[f"{name}_grad_x0"]) < 5e-6 and rel_err(gt, d[f"{name}_grad_theta"]) < 5e-6


@pytest.mark.parametrize("name", list(CASES))
def test_torch_loop_matches_reference(name):
    from viforsdes_amd.core.euler_maruyama import euler_maruyama
    d = _fx()
    pos, (horizon, dt) = list(d[f"{name}_pos"]), (float(v) for v in d[f"{name}_cfg"])
    th = torch.from_numpy(d[f"{name}_theta"]).requires_grad_(True)
    x0 = torch.from_numpy(d[f"{name}_x0"]).requires_grad_(True)
    tr = euler_maruyama(_sde(name), x0, th, horizon, dt, pos, noise=torch.from_numpy(d[f"{name}_noise"]))
    gt, gx = torch.autograd.grad((tr * torch.from_numpy(d[f"{name}_g_traj"])).sum(), [th, x0])
    assert rel_err(tr.detach().numpy(), d[f"{name}_traj"]) < 1e-6
    assert rel_err(gt.numpy(), d[f"{name}_grad_theta"]) < 1e-5 and rel_err(gx.numpy(), d[f"{name}_grad_x0"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_simulator_matches_reference(name):
    from viforsdes_amd.core import euler_maruyama as em
    d = _fx()
    dev = "cuda:0"
    pos, (horizon, dt) = list(d[f"{name}_pos"]), (float(v) for v in d[f"{name}_cfg"])
    th = torch.from_numpy(d[f"{name}_theta"]).to(dev).requires_grad_(True)
    x0 = torch.from_numpy(d[f"{name}_x0"]).to(dev).requires_grad_(True)
    tr = em.euler_maruyama(_sde(name), x0, th, horizon, dt, pos, noise=torch.from_numpy(d[f"{name}_noise"]).to(dev))
    assert tr.grad_fn is not None and "BuiltinEulerMaruyama" in type(tr.grad_fn).__name__, "the HIP simulator must be the path"
    gt, gx = torch.autograd.grad((tr * torch.from_numpy(d[f"{name}_g_traj"]).to(dev)).sum(), [th, x0])
    assert rel_err(tr.detach().cpu().numpy(), d[f"{name}_traj"]) < 2e-6
    if pos:  # the clamp fired in the fixture and must fire identically here
        assert int((tr[:, 1:] == 1e-6).sum()) == int((d[f"{name}_traj"][:, 1:] == np.float32(1e-6)).sum()) > 0
    assert rel_err(gt.cpu().numpy(), d[f"{name}_grad_theta"]) < 2e-5 and rel_err(gx.cpu().numpy(), d[f"{name}_grad_x0"]) < 2e-5


@pytest.mark.gpu
def test_hip_simulator_pretraining_size_vs_f64_oracle():
    """B=4096 Lotka-Volterra paths (what pretrain_sde_parameters simulates per iteration) against the float64 oracle: the
    forward over the full T=400 steps; the backward over the first 100 steps -- the adjoint of an oscillating LV path grows
    by orders of magnitude over 400 steps (|d traj / d theta| ~ 5e7 here), so beyond ~100 steps an fp32 adjoint (the
    reference's torch autograd included) agrees with a float64 one only in its leading digits."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    g = torch.Generator().manual_seed(8)
    B, T, dt = 4096, 400, 0.1
    # around the classical predator-prey parameters: oscillating paths, a few of which touch the 1e-6 floor
    theta = torch.tensor([0.5, 0.0025, 0.3]) * (1.0 + 0.1 * torch.rand(B, 3, generator=g))
    x0 = torch.tensor([[71.0, 79.0]]).expand(B, 2).contiguous()
    noise, gw = torch.randn(B, T, 2, generator=g), torch.randn(B, T + 1, 2, generator=g)
    d = lambda t: t.to("cuda:0")
    tr = _hip.euler_maruyama_fwd("lotka_volterra", d(x0), d(theta), d(noise), dt, [0, 1])
    sub = slice(0, 256)
    ref = vo.euler_maruyama("lv", x0[sub].numpy(), theta[sub].numpy(), noise[sub].numpy(), dt, [0, 1], np.float64)
    e_fwd = rel_err(tr[sub].cpu().numpy(), ref)
    Tb = 100
    trb = _hip.euler_maruyama_fwd("lotka_volterra", d(x0), d(theta), d(noise[:, :Tb].contiguous()), dt, [0, 1])
    gx, gt = _hip.euler_maruyama_bwd("lotka_volterra", d(theta), d(noise[:, :Tb].contiguous()), trb, d(gw[:, :Tb + 1].contiguous()), dt, [0, 1])
    rgx, rgt = vo.euler_maruyama_bwd("lv", theta[sub].numpy(), noise[sub, :Tb].numpy(), trb[sub].cpu().numpy(), gw[sub, :Tb + 1].numpy(),
                                     dt, [0, 1], np.float64)   # backward on the kernel's own trajectory (same clamp pattern)
    e = (e_fwd, rel_err(gx[sub].cpu().numpy(), rgx), rel_err(gt[sub].cpu().numpy(), rgt))
    print("\nEM LV B=4096: forward T=400, backward T=100 vs f64:", e)
    assert np.isfinite(tr.cpu().numpy()).all() and e[0] < 2e-4 and e[1] < 2e-4 and e[2] < 2e-4   # fp32 vs f64 over 400 oscillating steps: ~6e-5


@pytest.mark.gpu
def test_hip_simulator_linear_diagonal_matches_torch_loop():
    from viforsdes_amd.core import euler_maruyama as em
    from viforsdes_amd.examples.sdes import LinearDiagonalSDE
    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    S, B, horizon, dt = 8, 37, 1.0, 0.01
    sde = LinearDiagonalSDE(S)
    th = torch.randn(B, 2 * S, generator=g).to(dev)
    x0 = torch.randn(B, S, generator=g).to(dev)
    noise, gw = torch.randn(B, 100, S, generator=g).to(dev), torch.randn(B, 101, S, generator=g).to(dev)
    outs = []
    for hip in (True, False):
        em.HIP_SIMULATOR = hip
        try:
            t, x = th.clone().requires_grad_(True), x0.clone().requires_grad_(True)
            tr = em.euler_maruyama(sde, x, t, horizon, dt, [1, 5], noise=noise)
            outs.append((tr.detach(), *torch.autograd.grad((tr * gw).sum(), [t, x])))
        finally:
            em.HIP_SIMULATOR = True
    for a, b in zip(*outs):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5


@pytest.mark.gpu
def test_hip_simulator_propagates_nan_like_the_torch_loop():
    """A path whose theta draw is NaN (or overflows to inf - inf) must come out NaN on the positive dims too, exactly as
    ``torch.maximum`` / ``clamp`` do in the reference loop (core/euler_maruyama.py:38-42): ``fmaxf(NaN, 1e-6)`` would hide it
    from the non-finite-loss guard of the pre-training loop (trainer.py:240-247)."""
    from viforsdes_amd.core import euler_maruyama as em
    from viforsdes_amd.examples.sdes import LotkaVolterra, OrnsteinUhlenbeck
    from viforsdes_amd.inference.evidence_lower_bound import sde_coefficients
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    for sde, S, pos in ((LotkaVolterra(), 2, [0, 1]), (OrnsteinUhlenbeck(), 1, [0])):
        B, T = 6, 12
        theta = (0.1 + torch.rand(B, 3, generator=g)).to(dev)
        theta[1, 0] = float("nan")
        theta[4, 2] = float("inf")
        x0 = (1.0 + torch.rand(B, S, generator=g)).to(dev)
        noise = torch.randn(B, T, S, generator=g).to(dev)
        outs = []
        for hip in (True, False):
            em.HIP_SIMULATOR = hip
            try:
                outs.append(em.euler_maruyama(sde, x0, theta, T * 0.1, 0.1, pos, noise=noise))
            finally:
                em.HIP_SIMULATOR = True
        k, t = outs
        assert torch.equal(torch.isnan(k), torch.isnan(t)) and bool(torch.isnan(k[1, 1:]).all())
        ok = ~torch.isnan(t) & ~torch.isinf(t)
        assert rel_err(k[ok].cpu().numpy(), t[ok].cpu().numpy()) < 2e-5
        # the ELBO's coefficient kernel: NaN state or parameter -> NaN coefficients, as the Python callables give
        xs = k.clone()
        drift, diff = sde_coefficients(sde, xs, theta)
        xf = xs[:, :-1].reshape(B * T, S)
        tf = theta.unsqueeze(1).expand(B, T, -1).reshape(B * T, -1)
        assert torch.equal(torch.isnan(drift.reshape(B * T, S)), torch.isnan(sde.drift(xf, tf)))
        assert torch.equal(torch.isnan(diff.reshape(B * T, S, S)), torch.isnan(sde.diffusion(xf, tf)))
