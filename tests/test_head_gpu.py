"""GPU parity: fused HIP head (through the C ABI) vs the reference's golden vectors and the oracle.

Tolerances (fp32 kernels, v_exp/v_rcp based sigmoid/tanh): relative-to-max error
  forward  <= 2e-5  vs the float64 golden ("truth"), gradients <= 2e-4.
The fp32 reference itself sits at ~1e-6 / ~6e-6 from that truth on these cases.
"""
import numpy as np
import pytest
import torch

from helpers import G_NAMES, HEAD_CASES, MP_CASES, W_NAMES, load_head_case, rel_err

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
BWD_TOL = 2e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _run_case(d, bf16_ctx=False):
    from viforsdes_amd import _hip
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ws = [t(d["w_" + n].astype(np.float32)) for n in W_NAMES]
    ctx_full = t(d["context_full"])
    if bf16_ctx:
        ctx_full = ctx_full.to(torch.bfloat16)
    ctx = ctx_full[:, :-1]  # the reference's non-contiguous slice
    x0, theta, eps = t(d["x0"]), t(d["sde_parameters"]), t(d["eps"])
    dt = float(d["dt"])
    paths, means, chol, chol_raw, acts = _hip.head_forward(x0, ctx, theta, eps, ws, dt, True)
    grads = _hip.head_backward(t(d["g_paths"]), t(d["g_means"]), t(d["g_chol"]), ctx, theta, eps,
                               paths, chol_raw, acts, ws, dt)
    ev = _hip.head_forward(x0, ctx, theta, eps, ws, dt, False)
    torch.cuda.synchronize()
    return (paths, means, chol, chol_raw, acts), grads, ev


@pytest.mark.parametrize("name", HEAD_CASES)
def test_forward_backward_vs_golden(name):
    d = load_head_case(name)
    tag = "o1f64" if "o1f64_paths" in d else "o1f32"
    (paths, means, chol, chol_raw, acts), grads, ev = _run_case(d)
    for k, v in (("paths", paths), ("means", means), ("chol", chol)):
        assert rel_err(v.cpu().numpy(), d[f"{tag}_{k}"]) < FWD_TOL, k
    # eval (no-grad) launch gives the same numbers as the training launch
    for a, b_ in zip(ev[:3], (paths, means, chol)):
        assert torch.equal(a, b_)
    assert ev[3] is None and ev[4] is None
    # strict upper triangle of the Cholesky factor is exactly zero (reference forward.py:404-406)
    S = d["S"]
    iu = torch.triu_indices(S, S, 1)
    assert (chol[..., iu[0], iu[1]] == 0).all()
    for gname, g in zip(G_NAMES, grads):
        ref = d[f"{tag}_grad_{gname}"]
        if ref.size == 0:
            assert g.numel() == 0
            continue
        assert rel_err(g.cpu().numpy(), ref) < BWD_TOL, gname


@pytest.mark.parametrize("name", MP_CASES)
def test_default_dispatch_above_the_batch_thresholds_vs_golden(name):
    """Reference-generated cases (eager head definition looped, f64) with 300 / 700 paths: under the DEFAULT dispatch the forward
    (both cases) and the reverse-time sweep (700 paths) run the multi-path MFMA kernels; same tolerances as every other case."""
    test_forward_backward_vs_golden(name)


@pytest.mark.parametrize("np_group", [2, 4, 8, 16])
@pytest.mark.parametrize("name", ["ou_dims", "lv_dims"])
def test_multi_path_mfma_forward_vs_golden(name, np_group):
    """The reference-generated cases with hidden_dim 64 / two layers / state_dim 1, 2 through the multi-path MFMA forward kernel
    (csrc/vsde_head_mp.hip, forced, 2 / 4 / 8 / 16 paths per workgroup; batches of 3 / 2 paths = one partly filled group): same
    tolerances as the v2 kernel, the backward consumes the activations this forward saved."""
    from viforsdes_amd import _hip
    _hip.debug_head_mp(np_group)
    try:
        test_forward_backward_vs_golden(name)
        d = load_head_case(name)
        (paths, means, chol, chol_raw, acts), _, _ = _run_case(d)
    finally:
        _hip.debug_head_mp(-1)
    (p2, m2, c2, r2, a2), _, _ = _run_case(d)          # default route at this batch size: the v2 kernel
    for a, b_ in ((paths, p2), (means, m2), (chol, c2), (chol_raw, r2), (acts, a2)):
        assert rel_err(a.cpu().numpy(), b_.cpu().numpy()) < 5e-6


@pytest.mark.parametrize("name", ["tiny_l2", "clamp", "lv_dims"])
def test_matches_oracle_including_saved_activations(name):
    from oracle import vsde_oracle as vo
    d = load_head_case(name)
    (paths, means, chol, chol_raw, acts), grads, _ = _run_case(d)
    w = vo.HeadWeights(*[d["w_" + n] for n in W_NAMES])
    f = vo.head_forward(d["x0"], d["context_full"][:, :-1], d["sde_parameters"], d["eps"], w, float(d["dt"]), True)
    assert rel_err(acts.cpu().numpy(), f.acts) < FWD_TOL
    assert rel_err(chol_raw.cpu().numpy(), f.chol_raw) < FWD_TOL
    g = vo.head_backward(d["g_paths"], d["g_means"], d["g_chol"], d["context_full"][:, :-1], d["sde_parameters"],
                         d["eps"], f, w, float(d["dt"]))
    for gname, a, b_ in zip(G_NAMES, grads, g):
        if b_.size:
            assert rel_err(a.cpu().numpy(), b_) < BWD_TOL, gname


def test_bf16_context_is_read_directly():
    """A bf16 context (autocast) must give the same result as its fp32 up-cast."""
    from viforsdes_amd import _hip
    d = load_head_case("lv_dims")
    dev = _dev()
    (p16, m16, c16, _, _), g16, _ = _run_case(d, bf16_ctx=True)
    d2 = dict(d)
    d2["context_full"] = torch.from_numpy(d["context_full"]).to(torch.bfloat16).float().numpy()
    (p32, m32, c32, _, _), g32, _ = _run_case(d2)
    # the bf16 context takes the bf16-plane projection kernel (vsde_proj.hip: exact products, fp32 accumulation in a different
    # order than the fp32-MFMA kernel the up-cast copy takes), so the two runs agree to fp32 round-off, not bit for bit
    assert rel_err(p16.cpu().numpy(), p32.cpu().numpy()) < 2e-6
    for a, b_ in zip(g16, g32):
        if b_.numel():
            assert rel_err(a.float().cpu().numpy(), b_.float().cpu().numpy()) < 1e-5


def test_argument_errors_raise():
    from viforsdes_amd import _hip
    d = load_head_case("tiny_l2")
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ws = [t(d["w_" + n].astype(np.float32)) for n in W_NAMES]
    with pytest.raises(_hip.HipLibraryError):  # CPU tensors: no fallback
        _hip.head_forward(torch.zeros(2, 2), torch.zeros(2, 3, 16), torch.zeros(2, 3), torch.zeros(2, 3, 2),
                          [w.cpu() for w in ws], 0.1, False)
    bad = list(ws)
    bad[4] = torch.zeros(4, 24, 8, device=dev); bad[5] = torch.zeros(4, 24, 8, device=dev)  # 5 layers
    bad[6] = torch.zeros(4, 24, device=dev); bad[7] = torch.zeros(4, 24, device=dev)
    with pytest.raises(ValueError):
        _hip.head_forward(t(d["x0"]), t(d["context_full"])[:, :-1], t(d["sde_parameters"]), t(d["eps"]), bad,
                          float(d["dt"]), False)


@pytest.fixture()
def forward_kernel(request):
    """Forces the forward time-stepping kernel: 0 = four waves per path (v2), 2 / 4 / 8 / 16 = the multi-path MFMA kernel with that
    many paths per workgroup (the default picks by batch size, so a sub-batch could take another kernel than the full batch)."""
    from viforsdes_amd import _hip
    _hip.debug_head_mp(request.param)
    yield request.param
    _hip.debug_head_mp(-1)


@pytest.mark.parametrize("forward_kernel", [0, 2, 4, 8, 16], indirect=True, ids=["v2", "mfma2", "mfma4", "mfma8", "mfma16"])
def test_full_size_properties_lv(forward_kernel):
    """LV size (B=512, T=400, S=2, C=256, H=64, L=2): size-independent properties.

    * each sample path only depends on its own inputs -> running a sub-batch reproduces it;
    * the Euler-Maruyama identity z_{t+1} = z_t + mu dt + L eps sqrt(dt) holds on the outputs;
    * backward is linear in the upstream gradients and deterministic run-to-run.
    """
    from viforsdes_amd import _hip
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(3)
    B, T, S, C, P, H, L = 512, 400, 2, 256, 3, 64, 2
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(S + 3, H, sc=0.1), torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0]).to(dev)]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C).to(torch.bfloat16)[:, :-1], rn(B, P).abs(), rn(B, T, S)
    dt = 0.1
    paths, means, chol, chol_raw, acts = _hip.head_forward(x0, ctx, theta, eps, ws, dt, True)
    step = paths[:, :-1] + means * dt + torch.einsum("btij,btj->bti", chol, eps) * dt ** 0.5
    assert torch.allclose(step, paths[:, 1:], rtol=1e-5, atol=1e-5)
    assert (chol[..., 0, 0] >= 0.01).all() and (chol[..., 1, 1] >= 0.01).all() and (chol[..., 0, 1] == 0).all()
    sub = slice(100, 164)
    p2, m2, c2, _, _ = _hip.head_forward(x0[sub], ctx[sub], theta[sub], eps[sub], ws, dt, False)
    assert torch.equal(p2, paths[sub]) and torch.equal(m2, means[sub]) and torch.equal(c2, chol[sub])
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    run = lambda a, b_, c_: _hip.head_backward(a, b_, c_, ctx, theta, eps, paths, chol_raw, acts, ws, dt)
    g1 = run(gp, gm, gl)
    g1b = run(gp, gm, gl)
    for a, b_ in zip(g1, g1b):
        assert torch.equal(a, b_), "backward must be deterministic (no atomics)"
    # linearity in upstream grads holds except through the clamp pass-through rule, which only
    # depends on the SIGN of dL on clamped diagonals: scale by a positive constant.
    g2 = run(2 * gp, 2 * gm, 2 * gl)
    for a, b_ in zip(g1, g2):
        assert torch.allclose(2 * a, b_, rtol=1e-4, atol=1e-4 * float(b_.abs().max()))
    assert all(torch.isfinite(t_).all() for t_ in g1)


@pytest.mark.parametrize("name", ["tiny_l2", "ou_dims", "lv_dims", "clamp"])
def test_lds_resident_v1_kernels_also_match(name):
    """L <= 2 normally runs the register-resident v2 kernels; the one-wave-per-path v1 kernels (which serve
    L = 3, 4 and state dims with more than 16 emission rows) must give the same numbers."""
    from viforsdes_amd import _hip
    d = load_head_case(name)
    tag = "o1f64" if "o1f64_paths" in d else "o1f32"
    _hip.debug_force_v1(True)
    try:
        (paths, means, chol, chol_raw, acts), grads, _ = _run_case(d)
    finally:
        _hip.debug_force_v1(False)
    for k, v in (("paths", paths), ("means", means), ("chol", chol)):
        assert rel_err(v.cpu().numpy(), d[f"{tag}_{k}"]) < FWD_TOL, k
    for gname, g in zip(G_NAMES, grads):
        ref = d[f"{tag}_grad_{gname}"]
        if ref.size:
            assert rel_err(g.cpu().numpy(), ref) < BWD_TOL, gname


def test_full_size_ou_matches_oracle():
    """OU size (B=128, T=100, S=1, C=256, H=64, L=2): the whole batch against the CPU oracle."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    B, T, S, C, P, H, L = 128, 100, 1, 256, 3, 64, 2
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(S + 1, H, sc=0.1), torch.tensor([0.0, 0.5])]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C), rn(B, P).abs(), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    d_ = lambda t: t.to(dev)
    out = _hip.head_forward(d_(x0), d_(ctx)[:, :-1], d_(theta), d_(eps), [d_(w) for w in ws], 0.05, True)
    grads = _hip.head_backward(d_(gp), d_(gm), d_(gl), d_(ctx)[:, :-1], d_(theta), d_(eps), out[0], out[3], out[4],
                               [d_(w) for w in ws], 0.05)
    w = vo.HeadWeights(*[t.numpy() for t in ws])
    f = vo.head_forward(x0.numpy(), ctx.numpy()[:, :-1], theta.numpy(), eps.numpy(), w, 0.05, True, np.float64)
    gref = vo.head_backward(gp.numpy(), gm.numpy(), gl.numpy(), ctx.numpy()[:, :-1], theta.numpy(), eps.numpy(), f, w, 0.05,
                            np.float64)
    assert rel_err(out[0].cpu().numpy(), f.paths) < FWD_TOL and rel_err(out[2].cpu().numpy(), f.chol) < FWD_TOL
    for a, b_, n in zip(grads, gref, G_NAMES):
        if b_.size:
            assert rel_err(a.cpu().numpy(), b_) < BWD_TOL, n


def test_bf16_grad_context_matches_the_rounded_fp32_result():
    """grad_context written as bf16 (the autocast encoder's dtype) comes from the two-plane bf16-MFMA kernel (vsde_proj.hip,
    relative error 2^-16 before the final rounding): against the fp32 kernel's result rounded to bf16 it may differ by one
    bf16 ulp on a few elements, never more."""
    from viforsdes_amd import _hip
    g = torch.Generator().manual_seed(9)
    B, T, S, C, P, H, L = 24, 40, 2, 256, 3, 64, 2
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(_dev())
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(S + 3, H, sc=0.1), rn(S + 3, sc=0.1)]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C).to(torch.bfloat16), rn(B, P).abs(), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    out = _hip.head_forward(x0, ctx[:, :-1], theta, eps, ws, 0.1, True)
    g32 = _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, out[0], out[3], out[4], ws, 0.1)
    gctx = torch.zeros(B, T + 1, C, device=_dev(), dtype=torch.bfloat16)
    g16 = _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, out[0], out[3], out[4], ws, 0.1, context_grad_out=gctx)
    want = g32[1].to(torch.bfloat16).float()
    got = gctx[:, :-1].float()
    # one bf16 ulp at |want|, plus the kernel's 2^-16 relative to the size of the summed terms (matters where they cancel)
    tol = want.abs() * 2.0 ** -7 + float(want.abs().max()) * 2.0 ** -14
    diff = (got - want).abs()
    assert bool((diff <= tol).all()), float((diff / tol).max())
    assert float(((got != want).float().mean())) < 0.02, "more than 2% of the elements moved by an ulp"
    assert float(gctx[:, -1].abs().max()) == 0.0   # the extra context row receives no gradient
    for a, b_ in zip(g16[:1] + g16[2:], g32[:1] + g32[2:]):   # every other gradient is untouched by the output dtype
        if b_.numel():
            assert torch.equal(a, b_)


@pytest.mark.parametrize("S,L,T", [(2, 2, 1), (2, 2, 15), (2, 2, 17), (2, 2, 33), (1, 2, 48), (1, 1, 19), (3, 2, 21), (2, 2, 400)])
def test_ragged_step_counts_with_staggered_chunks(S, L, T):
    """Every path starts with a first chunk of a different length (the workgroups' chunk grids are staggered, forward and
    backward): step counts around the chunk size, a single step, and the full LV length, B = 37 paths (all 16 phases), against
    the float64 oracle (forward.py:137-375, backward.py:208-624 restated in oracle/vsde_oracle_impl.h)."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    dev = _dev()
    B, C, P, H = 37, 32, 3, 64
    NO = S + S * (S + 1) // 2
    g = np.random.default_rng(1000 * S + 10 * L + T)
    rn = lambda *s, sc=1.0: (g.standard_normal(s) * sc).astype(np.float32)
    ws = [rn(3 * H, S + C + P, sc=.15), rn(3 * H, H, sc=.15), rn(3 * H, sc=.1), rn(3 * H, sc=.1), rn(L - 1, 3 * H, H, sc=.15),
          rn(L - 1, 3 * H, H, sc=.15), rn(L - 1, 3 * H, sc=.1), rn(L - 1, 3 * H, sc=.1), rn(NO, H, sc=.2), rn(NO, sc=.3)]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C), np.abs(rn(B, P)), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tws = [t(w) for w in ws]
    out = _hip.head_forward(t(x0), t(ctx)[:, :-1], t(theta), t(eps), tws, 0.05, True)
    grads = _hip.head_backward(t(gp), t(gm), t(gl), t(ctx)[:, :-1], t(theta), t(eps), out[0], out[3], out[4], tws, 0.05)
    ev = _hip.head_forward(t(x0), t(ctx)[:, :-1], t(theta), t(eps), tws, 0.05, False)
    w64 = vo.HeadWeights(*[w.astype(np.float64) for w in ws])
    f64 = lambda a: a.astype(np.float64)
    f = vo.head_forward(f64(x0), f64(ctx)[:, :-1], f64(theta), f64(eps), w64, 0.05, True, dtype=np.float64)
    for a, b_ in zip(out[:3], (f.paths, f.means, f.chol)):
        assert rel_err(a.cpu().numpy(), b_) < FWD_TOL
    assert rel_err(out[4].cpu().numpy(), f.acts) < FWD_TOL and rel_err(out[3].cpu().numpy(), f.chol_raw) < FWD_TOL
    for a, b_ in zip(ev[:3], out[:3]):
        assert torch.equal(a, b_)
    gref = vo.head_backward(f64(gp), f64(gm), f64(gl), f64(ctx)[:, :-1], f64(theta), f64(eps), f, w64, 0.05, dtype=np.float64)
    for name, a, b_ in zip(G_NAMES, grads, gref):
        if b_.size:
            assert rel_err(a.cpu().numpy(), b_) < BWD_TOL, name


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("forward_kernel", [2, 4, 8, 16], indirect=True, ids=["mfma2", "mfma4", "mfma8", "mfma16"])
def test_multi_path_kernels_on_very_short_grids(forward_kernel, T):
    """One, two and three Euler steps (the kernels prefetch one / two steps ahead, store outputs one step late and walk the reverse sweep
    two steps per loop trip): forward and all 13 gradients against the float64 oracle, 5 paths = one partly filled group, one emission
    diagonal below the floor (the clamp and its gradient rule fire)."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    dev = _dev()
    g = torch.Generator().manual_seed(40 + T)
    B, S, C, P, H, L = 5, 2, 16, 3, 64, 2
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    ws = [rn(3 * H, S + C + P, sc=0.2), rn(3 * H, H, sc=0.2), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1), rn(L - 1, 3 * H, H, sc=0.2),
          rn(L - 1, 3 * H, H, sc=0.2), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1), rn(S + 3, H, sc=0.2),
          torch.tensor([0.0, 0.0, 0.7, 0.1, 0.005])]      # second diagonal below DIAG_MIN: the clamp and its gradient rule fire
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C), rn(B, P).abs(), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    d_ = lambda t: t.to(dev)
    out = _hip.head_forward(d_(x0), d_(ctx)[:, :-1], d_(theta), d_(eps), [d_(w) for w in ws], 0.05, True)
    grads = _hip.head_backward(d_(gp), d_(gm), d_(gl), d_(ctx)[:, :-1], d_(theta), d_(eps), out[0], out[3], out[4],
                               [d_(w) for w in ws], 0.05)
    w = vo.HeadWeights(*[t.numpy() for t in ws])
    f = vo.head_forward(x0.numpy(), ctx.numpy()[:, :-1], theta.numpy(), eps.numpy(), w, 0.05, True, np.float64)
    gref = vo.head_backward(gp.numpy(), gm.numpy(), gl.numpy(), ctx.numpy()[:, :-1], theta.numpy(), eps.numpy(), f, w, 0.05, np.float64)
    for a, b_, n in zip(out, (f.paths, f.means, f.chol, f.chol_raw, f.acts), ("paths", "means", "chol", "chol_raw", "acts")):
        assert rel_err(a.cpu().numpy(), b_) < FWD_TOL, n
    for a, b_, n in zip(grads, gref, G_NAMES):
        if b_.size:
            assert rel_err(a.cpu().numpy(), b_) < BWD_TOL, n


@pytest.mark.parametrize("forward_kernel", [2, 4, 8, 16], indirect=True, ids=["mfma2", "mfma4", "mfma8", "mfma16"])
def test_a_non_finite_path_does_not_contaminate_its_group(forward_kernel):
    """The multi-path forward puts 4 / 8 / 16 paths into the columns of one MFMA operand.  A path whose noise turns NaN / inf at some step
    must turn NaN / inf itself from there on (the emission floor propagates NaN like ``torch.maximum``, reference primitives/bounds.py:10-31)
    and must not touch the other paths of its group: they reproduce a clean run bit for bit, saved activations included."""
    from viforsdes_amd import _hip
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    B, T, S, C, P, H, L = 21, 12, 2, 32, 3, 64, 2
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ws = [rn(3 * H, S + C + P, sc=0.1), rn(3 * H, H, sc=0.15), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1), rn(L - 1, 3 * H, H, sc=0.15),
          rn(L - 1, 3 * H, H, sc=0.15), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1), rn(S + 3, H, sc=0.1),
          torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0]).to(dev)]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C), rn(B, P).abs(), rn(B, T, S)
    clean = _hip.head_forward(x0, ctx[:, :-1], theta, eps, ws, 0.1, True)
    bad = eps.clone()
    bad[6, 4, 1] = float("nan")          # path 6 from step 4 on
    bad[13, 7, 0] = float("inf")         # path 13 from step 7 on
    out = _hip.head_forward(x0, ctx[:, :-1], theta, bad, ws, 0.1, True)
    torch.cuda.synchronize()
    others = [i for i in range(B) if i not in (6, 13)]
    for a, b_ in zip(out, clean):
        assert torch.equal(a[others], b_[others])
    # path 6: eps[4, 1] only enters the second state component (L is lower triangular): z_5 = (finite, NaN), everything after is NaN
    assert torch.equal(out[0][6, :5], clean[0][6, :5]) and torch.isnan(out[0][6, 5, 1]) and torch.isnan(out[0][6, 6:]).all()
    assert torch.isnan(out[1][6, 5:]).all()              # the NaN state feeds the next step's GRU input: means are NaN from step 5 on
    assert torch.equal(out[0][13, :8], clean[0][13, :8]) and not torch.isfinite(out[0][13, 8:]).any()


def test_a_weight_beyond_the_f16_range_is_loud_then_served_by_the_fp32_kernels():
    """A recurrent weight above 2.2e4 cannot be a pair of f16 MFMA operands.  Contract (include/vsde_hip.h:
    vsde_head_mfma_range_exceeded): the launch that meets it returns non-finite paths -- never silently wrong ones -- and raises the
    sticky flag; every later launch takes the fp32 kernels and matches the float64 oracle again.  The flag is cleared at the end (it is
    process-wide)."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(21)
    B, T, S, C, P, H, L = 64, 24, 2, 256, 3, 64, 2
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(S + 3, H, sc=0.1), torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0])]
    ws[1][5, 7] = 5.0e4   # W_hh of layer 0
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C), rn(B, P).abs(), rn(B, T, S)
    d_ = lambda t: t.to(dev)
    wd = [d_(w) for w in ws]
    assert not _hip.head_mfma_range_exceeded()
    try:
        first = _hip.head_forward(d_(x0), d_(ctx)[:, :-1], d_(theta), d_(eps), wd, 0.1, False)
        torch.cuda.synchronize()
        assert _hip.head_mfma_range_exceeded(), "the weight preparation did not flag the weight"
        assert not bool(torch.isfinite(first[0]).all()), "the MFMA launch must not return finite paths for an operand it cannot hold"
        again = _hip.head_forward(d_(x0), d_(ctx)[:, :-1], d_(theta), d_(eps), wd, 0.1, False)
        w = vo.HeadWeights(*[t.numpy() for t in ws])
        f = vo.head_forward(x0.numpy(), ctx.numpy()[:, :-1], theta.numpy(), eps.numpy(), w, 0.1, False, np.float64)
        assert rel_err(again[0].cpu().numpy(), f.paths) < FWD_TOL and rel_err(again[2].cpu().numpy(), f.chol) < FWD_TOL
    finally:
        _hip.head_mfma_range_exceeded(clear=True)
    assert not _hip.head_mfma_range_exceeded()


@pytest.mark.parametrize("ctx_dtype", [torch.float32, torch.bfloat16], ids=["ctx32", "ctx16"])
@pytest.mark.parametrize("B,T,S", [(1, 16, 2), (2, 16, 2), (3, 16, 1), (5, 16, 2), (16, 17, 2), (16, 31, 3), (48, 25, 2), (7, 64, 2), (9, 48, 1),
                                   (64, 18, 2)])
def test_weight_gradient_tiles_on_split_operands_at_every_block_count(B, T, S, ctx_dtype):
    """Shapes the fast weight-gradient path takes (hidden 64, B T a multiple of 16, T >= 16; csrc/vsde_tn_wide.hip): 1, 2, 3, 5 ... 16-row
    blocks per split (the last group of four blocks is masked), blocks that straddle two batch rows (T = 17, 25, 31, 18), fp32 and bf16
    context operands (six / three products per block): all 13 gradients against the float64 oracle, twice with equal bits."""
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    dev = _dev()
    C, P, H, L = 256, 3, 64, 2
    NO = S + S * (S + 1) // 2
    g = np.random.default_rng(7000 + 100 * B + T + S)
    rn = lambda *s, sc=1.0: (g.standard_normal(s) * sc).astype(np.float32)
    ws = [rn(3 * H, S + C + P, sc=.08), rn(3 * H, H, sc=.12), rn(3 * H, sc=.1), rn(3 * H, sc=.1), rn(L - 1, 3 * H, H, sc=.12),
          rn(L - 1, 3 * H, H, sc=.12), rn(L - 1, 3 * H, sc=.1), rn(L - 1, 3 * H, sc=.1), rn(NO, H, sc=.1), rn(NO, sc=.3)]
    ctx = torch.from_numpy(rn(B, T + 1, C)).to(ctx_dtype)
    x0, theta, eps = rn(B, S), np.abs(rn(B, P)), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tws, cd = [t(w) for w in ws], ctx.to(dev)
    out = _hip.head_forward(t(x0), cd[:, :-1], t(theta), t(eps), tws, 0.05, True)
    run = lambda: _hip.head_backward(t(gp), t(gm), t(gl), cd[:, :-1], t(theta), t(eps), out[0], out[3], out[4], tws, 0.05)
    grads, again = run(), run()
    for a, b_ in zip(grads, again):
        assert torch.equal(a, b_)
    c64 = ctx.float().numpy().astype(np.float64)
    w64 = vo.HeadWeights(*[w.astype(np.float64) for w in ws])
    f64 = lambda a: a.astype(np.float64)
    f = vo.head_forward(f64(x0), c64[:, :-1], f64(theta), f64(eps), w64, 0.05, True, dtype=np.float64)
    gref = vo.head_backward(f64(gp), f64(gm), f64(gl), c64[:, :-1], f64(theta), f64(eps), f, w64, 0.05, dtype=np.float64)
    for name, a, b_ in zip(G_NAMES, grads, gref):
        if b_.size and name != "context":
            assert rel_err(a.float().cpu().numpy(), b_) < BWD_TOL, name
