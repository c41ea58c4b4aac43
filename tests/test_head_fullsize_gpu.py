"""GPU parity of the fused head AT THE BENCHMARK SIZES against the float64 CPU oracle.

Two workloads: Lotka-Volterra (B=512 launched, T=400, S=2, P=3, C=256, H=64, L=2, bf16 context)
and the synthetic stress config (B=256 launched, T=1000, S=8, P=16, C=512, H=64, L=2).  The oracle scores a sub-batch
of the launched paths (each path is independent: forward.py:91-135 runs one program per path).

Each workload is checked two ways, because a T-step recurrence amplifies rounding:

* teacher-forced -- every step re-evaluated in float64 from the kernel's OWN history (z_t, h_{t-1}), which scores the
  per-step arithmetic of the forward with no amplification, and the reverse-time backward run by the float64 oracle on the
  kernel's OWN saved activations, which scores the backward arithmetic alone;
* free-running -- kernel vs the oracle's own float64 trajectory from the same inputs: includes the amplification of fp32
  rounding (incl. v_exp/v_rcp based sigmoid/tanh) through T steps, hence the looser, stated tolerances.

Tolerances are relative to the max magnitude of the compared tensor:
  teacher-forced forward 2e-5, backward 2e-4 (the small-case tolerances of test_head_gpu.py);
  free-running forward 5e-5, backward 2e-4 at both sizes (measured on MI355X: forward <= 1e-6, backward <= 2e-6 -- the
  GRU is contractive at these weight scales, so T = 400 / 1000 steps do not amplify the fp32 rounding).
"""
import numpy as np
import pytest
import torch

from helpers import G_NAMES, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

WORKLOADS = {
    # name: (B launched, sub-batch scored, T, S, C, P, H, L, dt, bf16 context, free-running fwd tol, bwd tol)
    "lv": (512, 64, 400, 2, 256, 3, 64, 2, 0.1, True, 5e-5, 2e-4),
    "synthetic": (256, 32, 1000, 8, 512, 16, 64, 2, 0.01, True, 5e-5, 2e-4),
    # small shapes that still take the hidden_dim-64 fast paths (192-wide weight-gradient tiles need B*T % 16 == 0 and T >= 16;
    # bf16-plane projection needs a 256-wide bf16 context): narrow state / theta tiles, swapped emission tile, 1 and 3 GRU layers,
    # fp32 context (generic projection + wide weight gradients), a context width that is a single 64-column tile
    "ou_like_l1": (12, 4, 48, 1, 256, 3, 64, 1, 0.05, True, 5e-5, 2e-4),
    "s3_l3_fp32ctx": (8, 4, 32, 3, 128, 5, 64, 3, 0.05, False, 5e-5, 2e-4),
    "c64_t16": (6, 2, 16, 2, 64, 3, 64, 2, 0.1, False, 5e-5, 2e-4),
    # emission rows wider than 16 (state dimensions 5..9): the swapped weight-gradient tile is split into 16-column pieces
    # (20 = 16 + 4 and 54 = 3 x 16 + 6 columns)
    "s5_l1": (8, 4, 32, 5, 128, 4, 64, 1, 0.05, False, 5e-5, 2e-4),
    "s9_l2": (8, 4, 32, 9, 64, 5, 64, 2, 0.05, True, 5e-5, 2e-4),
}
TF_FWD_TOL, TF_BWD_TOL = 2e-5, 2e-4


def _inputs(B, T, S, C, P, H, L, seed, bf16_ctx):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    NO = S + S * (S + 1) // 2
    bias = torch.zeros(NO)
    for k in range(S):
        bias[S + k * (k + 3) // 2] = 1.0          # default emission bias: unit diagonal (head.py:60-66)
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(NO, H, sc=0.1), bias]
    ctx = rn(B, T + 1, C)
    if bf16_ctx:
        ctx = ctx.to(torch.bfloat16)              # what the autocast encoder hands over; both sides read these values
    return ws, rn(B, S), ctx, rn(B, P).abs(), rn(B, T, S), rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)


# workloads the multi-path MFMA forward kernel (csrc/vsde_head_mp.hip: hidden_dim 64, L <= 2, state_dim <= 2) can take run twice,
# once per forward kernel (forced: the sub-batch launch below must take the same kernel as the full launch)
CASES = [(n, mp) for n, v in WORKLOADS.items() for mp in ((0, 2, 4, 16) if (v[3] <= 2 and v[7] <= 2 and v[6] == 64) else (0,))]


@pytest.fixture()
def forward_kernel(request):
    from viforsdes_amd import _hip
    _hip.debug_head_mp(request.param)
    yield request.param
    _hip.debug_head_mp(-1)


@pytest.mark.parametrize("name,forward_kernel", CASES, indirect=["forward_kernel"],
                         ids=[f"{n}-{'mfma%d' % mp if mp else 'v2'}" for n, mp in CASES])
def test_head_full_size_vs_f64_oracle(name, forward_kernel):
    from oracle import vsde_oracle as vo
    from viforsdes_amd import _hip
    B, Bs, T, S, C, P, H, L, dt, bf16_ctx, fr_fwd, fr_bwd = WORKLOADS[name]
    ws, x0, ctx_full, theta, eps, gp, gm, gl = _inputs(B, T, S, C, P, H, L, 100 + len(name), bf16_ctx)
    d = lambda t: t.to(DEV)
    wd = [d(w) for w in ws]
    ctx_d = d(ctx_full)[:, :-1]
    # full launch (the benchmark's grid) ...
    paths, means, chol, chol_raw, acts = _hip.head_forward(d(x0), ctx_d, d(theta), d(eps), wd, dt, True)
    grads = _hip.head_backward(d(gp), d(gm), d(gl), ctx_d, d(theta), d(eps), paths, chol_raw, acts, wd, dt)
    # ... and the scored sub-batch launched alone, for the weight gradients (sums over the launched paths)
    sub = slice(B // 4, B // 4 + Bs)
    fs = _hip.head_forward(d(x0)[sub], ctx_d[sub], d(theta)[sub], d(eps)[sub], wd, dt, True)
    gs = _hip.head_backward(d(gp)[sub], d(gm)[sub], d(gl)[sub], ctx_d[sub], d(theta)[sub], d(eps)[sub], fs[0], fs[3], fs[4],
                            wd, dt)
    torch.cuda.synchronize()
    for a, b_ in zip(fs[:3], (paths, means, chol)):
        assert torch.equal(a, b_[sub]), "a path must not depend on which launch it is part of"
    for i in (0, 1, 2):  # per-path gradients: x0, context, theta
        assert torch.equal(gs[i], grads[i][sub])

    n = lambda t: t.detach().cpu().numpy()
    w = vo.HeadWeights(*[n(t) for t in ws])
    ctx_np = n(ctx_full.float())[sub, :-1]
    x0n, thn, epn = n(x0)[sub], n(theta)[sub], n(eps)[sub]
    gpn, gmn, gln = n(gp)[sub], n(gm)[sub], n(gl)[sub]
    k_paths, k_means, k_chol, k_raw, k_acts = (n(t)[sub] for t in (paths, means, chol, chol_raw, acts))

    # ---- teacher-forced forward: every step from the kernel's own (z_t, h_{t-1}) in float64
    a_ref, m_ref, raw_ref, nxt_ref = vo.head_steps_teacher_forced(k_paths, k_acts, ctx_np, thn, epn, w, dt)
    errs = {"tf_acts": rel_err(k_acts, a_ref), "tf_means": rel_err(k_means, m_ref), "tf_chol_raw": rel_err(k_raw, raw_ref),
            "tf_next_state": rel_err(k_paths[:, 1:], nxt_ref)}
    # ---- teacher-forced backward: the float64 oracle on the kernel's saved tensors
    saved = vo.FwdResult(k_paths, k_means, k_chol, k_raw, k_acts)
    g_tf = vo.head_backward(gpn, gmn, gln, ctx_np, thn, epn, saved, w, dt, np.float64)
    for gname, a, b_ in zip(G_NAMES, gs, g_tf):
        if b_.size:
            errs["tf_grad_" + gname] = rel_err(n(a), b_)
    # ---- free-running: the oracle's own float64 trajectory
    f = vo.head_forward(x0n, ctx_np, thn, epn, w, dt, True, np.float64)
    g_fr = vo.head_backward(gpn, gmn, gln, ctx_np, thn, epn, f, w, dt, np.float64)
    errs.update(fr_paths=rel_err(k_paths, f.paths), fr_means=rel_err(k_means, f.means), fr_chol=rel_err(k_chol, f.chol))
    for gname, a, b_ in zip(G_NAMES, gs, g_fr):
        if b_.size:
            errs["fr_grad_" + gname] = rel_err(n(a), b_)
    print(f"\n[{name}] " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert np.isfinite(k_paths).all() and float(np.abs(f.paths).max()) < 1e6, "synthetic weights must give bounded paths"
    for k, v in errs.items():
        tol = (TF_FWD_TOL if "grad" not in k else TF_BWD_TOL) if k.startswith("tf_") else (fr_fwd if "grad" not in k else fr_bwd)
        assert v < tol, (k, v, tol)


@pytest.mark.ablation_build
def test_weight_gradient_forms_of_the_head_agree_at_full_size():
    """The split-operand bf16 tiles (default), the fp32-MFMA tiles and both workgroup orders of vsde_tn_wide.hip, each in its own process
    on the ablation library (the switches are read once per process): all 13 gradients of the LV-size reverse pass within fp32
    round-off of the fp32-MFMA form (the measured figure is 1e-6 of the peak gradient; profiles/r06_tn_wide_split.txt)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "tn_wide_check.py")], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("VSDE_TW_SPLIT")]
    assert r.returncode == 0 and len(lines) == 4, r.stdout + r.stderr
    for ln in lines:
        mt = re.search(r": ok, max relative-to-peak difference from the first form ([0-9.e+-]+)", ln)
        assert mt, ln
        assert float(mt.group(1)) <= (0.0 if "SPLIT=0" in ln else 5e-6), ln
