"""CPU: pin the oracle (oracle/vsde_oracle_impl.h) against the golden vectors generated from the
reference (tests/golden/make_golden.py).  f64 oracle vs f64 reference must agree to ~1e-13;
f32 vs f32 to rounding."""
import numpy as np
import pytest

from helpers import G_NAMES, GOLDEN, HEAD_CASES, MP_CASES, W_NAMES, load_head_case, rel_err
from oracle import vsde_oracle as vo


def _weights(d):
    return vo.HeadWeights(*[d["w_" + n] for n in W_NAMES])


@pytest.mark.parametrize("name", HEAD_CASES + MP_CASES)
@pytest.mark.parametrize("dtype,tag,ftol,btol", [(np.float32, "o1f32", 5e-6, 2e-5), (np.float64, "o1f64", 1e-12, 1e-12)])
def test_head_forward_backward(name, dtype, tag, ftol, btol):
    d = load_head_case(name)
    if tag + "_paths" not in d:
        pytest.skip("case stored in f32 only")
    ctx = d["context_full"][:, :-1]
    f = vo.head_forward(d["x0"], ctx, d["sde_parameters"], d["eps"], _weights(d), float(d["dt"]), True, dtype)
    for k in ("paths", "means", "chol"):
        assert rel_err(getattr(f, k), d[f"{tag}_{k}"]) < ftol, k
    g = vo.head_backward(d["g_paths"], d["g_means"], d["g_chol"], ctx, d["sde_parameters"], d["eps"], f, _weights(d),
                         float(d["dt"]), dtype)
    for k in G_NAMES:
        ref = d[f"{tag}_grad_{k}"]
        if ref.size:
            assert rel_err(getattr(g, k), ref) < btol, k


@pytest.mark.parametrize("name", ["tiny_l2", "tiny_l1_odd"])
def test_oracle_matches_the_reference_triton_kernels(name):
    """O2 = the reference's sde_fwd_kernel/sde_bwd_kernel run under TRITON_INTERPRET=1."""
    d = load_head_case(name)
    ctx = d["context_full"][:, :-1]
    f = vo.head_forward(d["x0"], ctx, d["sde_parameters"], d["eps"], _weights(d), float(d["dt"]), True)
    for k in ("paths", "means", "chol"):
        assert rel_err(getattr(f, k), d["o2_" + k]) < 5e-6
    g = vo.head_backward(d["g_paths"], d["g_means"], d["g_chol"], ctx, d["sde_parameters"], d["eps"], f, _weights(d),
                         float(d["dt"]))
    for k in ("x0", "context", "sde_parameters", "W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0", "out_weight", "out_bias"):
        assert rel_err(getattr(g, k), d["o2_grad_" + k]) < 3e-5, k


def test_clamp_case_exercises_both_branches():
    d = load_head_case("clamp")
    f = vo.head_forward(d["x0"], d["context_full"][:, :-1], d["sde_parameters"], d["eps"], _weights(d), float(d["dt"]), True)
    S = d["S"]
    diag = [k * (k + 3) // 2 for k in range(S)]
    raw = f.chol_raw[..., diag]
    assert (raw < vo.DIAG_MIN).any() and (raw >= vo.DIAG_MIN).any()
    assert (np.diagonal(f.chol, axis1=-2, axis2=-1) >= vo.DIAG_MIN - 1e-9).all()


def test_strided_context_view_is_read_in_place():
    d = load_head_case("tiny_l2")
    full = d["context_full"]
    a = vo.head_forward(d["x0"], full[:, :-1], d["sde_parameters"], d["eps"], _weights(d), float(d["dt"]), False)
    b = vo.head_forward(d["x0"], np.ascontiguousarray(full[:, :-1]), d["sde_parameters"], d["eps"], _weights(d),
                        float(d["dt"]), False)
    assert np.array_equal(a.paths, b.paths)


@pytest.mark.parametrize("name", ["ou", "lv"])
@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-6), (np.float64, 2e-5)])  # f64 vs an f32 fixture
def test_elbo_terms(name, dtype, tol):
    d = dict(np.load(f"{GOLDEN}/elbo_{name}.npz"))
    B, T, S, P = (int(v) for v in d["dims"])
    dt = float(d["dt"])
    sp, tp = list(d["state_positive_dims"]), list(d["theta_positive_dims"])
    x = vo.to_state(d["z"], sp, dtype)
    assert rel_err(x, d["x"]) < 1e-6
    assert rel_err(vo.to_latent(d["x0"], sp, dtype), d["z0"]) < 1e-6
    s, g, j = vo.elbo_path_terms(d["z"], x, d["means"], d["chol"], d["drift"], d["diffusion"], sp, dt, dtype)
    o = vo.obs_log_prob(x, d["obs_idx"], d["obs_values"], float(d["variance"]), None, dtype)
    pr, po = vo.theta_log_probs(d["theta"], d["q_mean"], d["q_log_std"], tp, int(d["prior_type"]),
                                float(d["prior_mean"]), float(d["prior_std"]), dtype)
    assert rel_err(s, d["sde_lp"]) < tol and rel_err(g, d["gen_lp"]) < tol
    assert rel_err(o, d["obs_lp"]) < tol and rel_err(pr, d["prior_lp"]) < tol and rel_err(po, d["post_lp"]) < tol
    if d["jac"].any():
        assert rel_err(j, d["jac"]) < tol
    elbo = float((o + s - g + j + pr - po).mean())
    assert abs(elbo - float(d["elbo"])) <= 2e-6 * abs(float(d["elbo"]))
    one = np.full((B,), 1.0 / B)
    gz, gx, gm, gc, gf, gG = vo.elbo_path_terms_bwd(d["z"], x, d["means"], d["chol"], d["drift"], d["diffusion"], sp, dt,
                                                    one, -one, one, dtype)
    assert rel_err(gm, d["grad_means"]) < 10 * tol and rel_err(gc, d["grad_chol"]) < 10 * tol


def test_elbo_bwd_matches_finite_differences():
    rng = np.random.default_rng(0)
    B, T, S = 2, 3, 3
    z = rng.normal(size=(B, T + 1, S)); means = rng.normal(size=(B, T, S)); drift = rng.normal(size=(B, T, S))
    mk = lambda: np.tril(rng.normal(size=(B, T, S, S)) * 0.3, -1) + np.eye(S) * (0.5 + rng.random((B, T, S, 1)))
    chol, diff = mk(), mk()
    pos = [0, 2]
    x = vo.to_state(z, pos, np.float64)
    gs, gg, gj = rng.normal(size=B), rng.normal(size=B), rng.normal(size=B)

    def f(z_, x_, m_, c_, d_, G_):
        s, g, j = vo.elbo_path_terms(z_, x_, m_, c_, d_, G_, pos, 0.07, np.float64)
        return float((gs * s + gg * g + gj * j).sum())

    grads = vo.elbo_path_terms_bwd(z, x, means, chol, drift, diff, pos, 0.07, gs, gg, gj, np.float64)
    args = [z, x, means, chol, drift, diff]
    for ai, (arr, gr) in enumerate(zip(args, grads)):
        for _ in range(6):
            idx = tuple(rng.integers(0, n) for n in arr.shape)
            if ai in (3, 5) and idx[-1] > idx[-2]:
                continue  # strict upper triangle is not an input of the factor
            h = 1e-6
            ap, am = [a.copy() for a in args], [a.copy() for a in args]
            ap[ai][idx] += h; am[ai][idx] -= h
            fd = (f(*ap) - f(*am)) / (2 * h)
            assert abs(fd - gr[idx]) < 1e-5 * max(1.0, abs(fd)), (ai, idx, fd, gr[idx])
