"""TEST INFRASTRUCTURE ONLY -- numpy/ctypes front end of the CPU oracle.

The arithmetic lives in ``vsde_oracle_impl.h`` (plain C, each function cites the
reference lines it restates).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the shipped package
(``viforsdes_amd``) never does and fails loudly when its HIP library is missing.

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against the
golden vectors in ``tests/golden/`` that were produced by importing the reference
itself (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvsde_oracle.so")
DIAG_MIN = 1e-2  # reference: inference/constants.py:6


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("vsde_oracle.c", "vsde_oracle_impl.h")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-B", "libvsde_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


class HeadWeights(NamedTuple):
    """nn.GRU-native layout, as passed to ``_SDEFunction.apply`` (kernels/autograd.py:46-56)."""

    W_ih_l0: np.ndarray    # [3H, S+C+P]
    W_hh_l0: np.ndarray    # [3H, H]
    b_ih_l0: np.ndarray    # [3H]
    b_hh_l0: np.ndarray    # [3H]
    W_ih_stack: np.ndarray  # [L-1, 3H, H]
    W_hh_stack: np.ndarray  # [L-1, 3H, H]
    b_ih_stack: np.ndarray  # [L-1, 3H]
    b_hh_stack: np.ndarray  # [L-1, 3H]
    out_weight: np.ndarray  # [S+ntril, H]
    out_bias: np.ndarray    # [S+ntril]


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _dims(x0, ctx, theta, w: HeadWeights):
    B, S = x0.shape
    T, C = ctx.shape[1], ctx.shape[2]
    P = theta.shape[1]
    H = w.W_hh_l0.shape[1]
    L = 1 + (w.W_ih_stack.shape[0] if w.W_ih_stack.size else 0)
    assert w.W_ih_l0.shape == (3 * H, S + C + P), (w.W_ih_l0.shape, (3 * H, S + C + P))
    return B, T, S, P, C, H, L


def _ctx_arg(ctx: np.ndarray, dtype):
    """Accept the strided ``context[:, :-1]`` view without copying when possible."""
    item = np.dtype(dtype).itemsize
    if ctx.dtype == dtype and ctx.strides[2] == item and ctx.strides[1] == ctx.shape[2] * item \
            and ctx.strides[0] % item == 0:
        return ctx, ctx.strides[0] // item
    c = _c(ctx, dtype)
    return c, c.shape[1] * c.shape[2]


def _wargs(w: HeadWeights, dtype):
    return [_c(a, dtype) for a in w]


class FwdResult(NamedTuple):
    paths: np.ndarray       # [B, T+1, S]
    means: np.ndarray       # [B, T, S]
    chol: np.ndarray        # [B, T, S, S]
    chol_raw: Optional[np.ndarray]  # [B, T, ntril]
    acts: Optional[np.ndarray]      # [B, T, L, 5, H]  (h, r, u, n, c_n)


def head_forward(x0, ctx, theta, eps, w: HeadWeights, dt: float, save: bool = True,
                 dtype=np.float32, diag_min: float = DIAG_MIN) -> FwdResult:
    """launch_fwd restated (kernels/forward.py:378-563)."""
    sfx, _ = _dt(dtype)
    x0 = _c(x0, dtype); theta = _c(theta, dtype); eps = _c(eps, dtype)
    ctx, bstride = _ctx_arg(np.asarray(ctx), dtype)
    B, T, S, P, C, H, L = _dims(x0, ctx, theta, w)
    ntril = S * (S + 1) // 2
    ws = _wargs(w, dtype)
    paths = np.empty((B, T + 1, S), dtype)
    means = np.empty((B, T, S), dtype)
    chol = np.empty((B, T, S, S), dtype)
    chol_raw = np.empty((B, T, ntril), dtype) if save else None
    acts = np.empty((B, T, L, 5, H), dtype) if save else None
    fn = getattr(lib(), "vsde_oracle_fwd" + sfx)
    fn.restype = None
    fn(*(ctypes.c_int(v) for v in (B, T, S, P, C, H, L)),
       _p(x0), _p(ctx), ctypes.c_long(bstride), _p(theta), _p(eps),
       *(_p(a) for a in ws), ctypes.c_double(dt), ctypes.c_double(diag_min),
       _p(paths), _p(means), _p(chol), _p(chol_raw), _p(acts))
    return FwdResult(paths, means, chol, chol_raw, acts)


class BwdResult(NamedTuple):
    """Same order as launch_bwd's 13-tuple (kernels/backward.py:766-784)."""

    x0: np.ndarray
    context: np.ndarray
    sde_parameters: np.ndarray
    W_ih_l0: np.ndarray
    W_hh_l0: np.ndarray
    b_ih_l0: np.ndarray
    b_hh_l0: np.ndarray
    W_ih_stack: np.ndarray
    W_hh_stack: np.ndarray
    b_ih_stack: np.ndarray
    b_hh_stack: np.ndarray
    out_weight: np.ndarray
    out_bias: np.ndarray


def head_backward(g_paths, g_means, g_chol, ctx, theta, eps, fwd: FwdResult, w: HeadWeights,
                  dt: float, dtype=np.float32, diag_min: float = DIAG_MIN) -> BwdResult:
    """launch_bwd restated (kernels/backward.py:627-784)."""
    sfx, _ = _dt(dtype)
    theta = _c(theta, dtype); eps = _c(eps, dtype)
    ctx, bstride = _ctx_arg(np.asarray(ctx), dtype)
    B, T, C = ctx.shape
    S = fwd.paths.shape[2]
    P = theta.shape[1]
    H = w.W_hh_l0.shape[1]
    L = fwd.acts.shape[2]
    ntril = S * (S + 1) // 2
    ws = _wargs(w, dtype)
    g = BwdResult(
        np.empty((B, S), dtype), np.empty((B, T, C), dtype), np.empty((B, P), dtype),
        np.empty((3 * H, S + C + P), dtype), np.empty((3 * H, H), dtype),
        np.empty((3 * H,), dtype), np.empty((3 * H,), dtype),
        np.zeros((L - 1, 3 * H, H), dtype), np.zeros((L - 1, 3 * H, H), dtype),
        np.zeros((L - 1, 3 * H), dtype), np.zeros((L - 1, 3 * H), dtype),
        np.empty((S + ntril, H), dtype), np.empty((S + ntril,), dtype))
    fn = getattr(lib(), "vsde_oracle_bwd" + sfx)
    fn.restype = None
    fn(*(ctypes.c_int(v) for v in (B, T, S, P, C, H, L)),
       _p(_c(g_paths, dtype)), _p(_c(g_means, dtype)), _p(_c(g_chol, dtype)),
       _p(ctx), ctypes.c_long(bstride), _p(theta), _p(eps),
       _p(_c(fwd.paths, dtype)), _p(_c(fwd.chol_raw, dtype)), _p(_c(fwd.acts, dtype)),
       _p(ws[0]), _p(ws[1]), _p(ws[4]), _p(ws[5]), _p(ws[8]),
       ctypes.c_double(dt), ctypes.c_double(diag_min),
       *(_p(a) for a in g))
    return g


def _mask(pos_dims, n):
    m = np.zeros((n,), np.uint8)
    for d in pos_dims or []:
        m[d] = 1
    return m


def to_state(z, positive_dims, dtype=np.float32):
    sfx, _ = _dt(dtype)
    z = _c(z, dtype)
    S = z.shape[-1]
    x = np.empty_like(z)
    fn = getattr(lib(), "vsde_oracle_to_state" + sfx); fn.restype = None
    fn(ctypes.c_long(z.size // S), ctypes.c_int(S), _p(z), _p(_mask(positive_dims, S)), _p(x))
    return x


def to_latent(x, positive_dims, dtype=np.float32):
    sfx, _ = _dt(dtype)
    x = _c(x, dtype)
    S = x.shape[-1]
    z = np.empty_like(x)
    fn = getattr(lib(), "vsde_oracle_to_latent" + sfx); fn.restype = None
    fn(ctypes.c_long(x.size // S), ctypes.c_int(S), _p(x), _p(_mask(positive_dims, S)), _p(z))
    return z


def elbo_path_terms(z, x, means, chol, drift, diffusion, positive_dims, dt, dtype=np.float32):
    """(sde_log_prob[B], gen_log_prob[B], log_jacobian[B]) -- evidence_lower_bound.py:42-50."""
    sfx, _ = _dt(dtype)
    z = _c(z, dtype); x = _c(x, dtype); means = _c(means, dtype); chol = _c(chol, dtype)
    drift = _c(drift, dtype); diffusion = _c(diffusion, dtype)
    B, T1, S = z.shape
    T = T1 - 1
    assert S <= 64
    out = [np.empty((B,), dtype) for _ in range(3)]
    fn = getattr(lib(), "vsde_oracle_elbo_path_terms" + sfx); fn.restype = None
    fn(ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S), _p(z), _p(x), _p(means), _p(chol),
       _p(drift), _p(diffusion), _p(_mask(positive_dims, S)), ctypes.c_double(dt), *(_p(o) for o in out))
    return tuple(out)


def elbo_path_terms_bwd(z, x, means, chol, drift, diffusion, positive_dims, dt,
                        g_sde, g_gen, g_jac, dtype=np.float32):
    """Gradients (z, x, means, chol, drift, diffusion) of the three path terms."""
    sfx, _ = _dt(dtype)
    z = _c(z, dtype); x = _c(x, dtype); means = _c(means, dtype); chol = _c(chol, dtype)
    drift = _c(drift, dtype); diffusion = _c(diffusion, dtype)
    B, T1, S = z.shape
    T = T1 - 1
    outs = [np.zeros_like(z), np.zeros_like(x), np.zeros_like(means), np.zeros_like(chol),
            np.zeros_like(drift), np.zeros_like(diffusion)]
    fn = getattr(lib(), "vsde_oracle_elbo_path_terms_bwd" + sfx); fn.restype = None
    fn(ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S), _p(z), _p(x), _p(means), _p(chol),
       _p(drift), _p(diffusion), _p(_mask(positive_dims, S)), ctypes.c_double(dt),
       _p(_c(g_sde, dtype)), _p(_c(g_gen, dtype)), _p(_c(g_jac, dtype)), *(_p(o) for o in outs))
    return tuple(outs)


def obs_log_prob(x, obs_idx, obs_values, variance, obs_matrix=None, dtype=np.float32):
    sfx, _ = _dt(dtype)
    x = _c(x, dtype); obs_values = _c(obs_values, dtype)
    obs_idx = np.ascontiguousarray(obs_idx, dtype=np.int64)
    B, T1, S = x.shape
    n_obs, obs_dim = obs_values.shape
    om = None if obs_matrix is None else _c(obs_matrix, dtype)
    out = np.empty((B,), dtype)
    fn = getattr(lib(), "vsde_oracle_obs_log_prob" + sfx); fn.restype = None
    fn(ctypes.c_int(B), ctypes.c_int(T1 - 1), ctypes.c_int(S), ctypes.c_int(n_obs), ctypes.c_int(obs_dim),
       _p(x), _p(obs_idx), _p(obs_values), _p(om), ctypes.c_double(variance), _p(out))
    return out


def theta_log_probs(theta, q_mean, q_log_std, positive_dims, prior_is_lognormal, prior_mean, prior_std,
                    dtype=np.float32):
    sfx, _ = _dt(dtype)
    theta = _c(theta, dtype)
    B, P = theta.shape
    prior_lp = np.empty((B,), dtype); post_lp = np.empty((B,), dtype)
    fn = getattr(lib(), "vsde_oracle_theta_log_probs" + sfx); fn.restype = None
    fn(ctypes.c_int(B), ctypes.c_int(P), _p(theta), _p(_c(q_mean, dtype)), _p(_c(q_log_std, dtype)),
       _p(_mask(positive_dims, P)), ctypes.c_int(int(bool(prior_is_lognormal))),
       ctypes.c_double(prior_mean), ctypes.c_double(prior_std), _p(prior_lp), _p(post_lp))
    return prior_lp, post_lp


def head_steps_teacher_forced(paths, acts, ctx, theta, eps, w: HeadWeights, dt: float, diag_min: float = DIAG_MIN):
    """One-step-ahead ("teacher-forced") float64 evaluation of the head, vectorised over (B, T).

    Step t of forward.py:195-365 is re-evaluated from a GIVEN history -- z_t = ``paths[:, t]`` and
    h^l_{t-1} = ``acts[:, t-1, l, 0]`` (zeros at t = 0, forward.py:143) -- instead of the oracle's own, so a
    kernel's per-step arithmetic can be scored without the error amplification of a free-running path.
    Returns (acts_ref [B,T,L,5,H], means_ref [B,T,S], chol_raw_ref [B,T,ntril], next_ref [B,T,S]) where
    ``next_ref[:, t]`` is z_{t+1} computed from z_t with the Euler-Maruyama update (forward.py:352-365)."""
    f8 = np.float64
    paths = np.asarray(paths, f8); acts = np.asarray(acts, f8); ctx = np.asarray(ctx, f8)
    theta = np.asarray(theta, f8); eps = np.asarray(eps, f8)
    W = [np.asarray(a, f8) for a in w]
    B, T, L, _, H = acts.shape
    S = paths.shape[2]
    C = ctx.shape[2]
    ntril = S * (S + 1) // 2
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    hprev = np.concatenate([np.zeros((B, 1, L, H)), acts[:, :-1, :, 0]], axis=1)      # h^l_{t-1}
    z = paths[:, :-1]
    Wi0, Wh0, bi0, bh0 = W[0], W[1], W[2], W[3]
    a = (bi0 + theta @ Wi0[:, S + C:].T)[:, None] + z @ Wi0[:, :S].T + ctx @ Wi0[:, S:S + C].T
    out = np.empty((B, T, L, 5, H))
    inp = None
    for l in range(L):
        if l > 0:
            a = W[6][l - 1] + inp @ W[4][l - 1].T
            c = W[7][l - 1] + hprev[:, :, l] @ W[5][l - 1].T
        else:
            c = bh0 + hprev[:, :, 0] @ Wh0.T
        r = sig(a[..., :H] + c[..., :H]); u = sig(a[..., H:2 * H] + c[..., H:2 * H])
        n = np.tanh(a[..., 2 * H:] + r * c[..., 2 * H:])
        h = (1.0 - u) * n + u * hprev[:, :, l]
        out[:, :, l, 0], out[:, :, l, 1], out[:, :, l, 2], out[:, :, l, 3], out[:, :, l, 4] = h, r, u, n, c[..., 2 * H:]
        inp = h
    o = W[9] + inp @ W[8].T
    means = o[..., :S]
    raw = o[..., S:]
    Lm = np.zeros((B, T, S, S))
    ii, jj = np.tril_indices(S)                                                     # row-major tril order, head.py:88-97
    Lm[..., ii, jj] = raw
    d = np.arange(S)
    Lm[..., d, d] = np.maximum(Lm[..., d, d], diag_min)                            # forward.py:346-351
    nxt = z + means * dt + np.einsum("btij,btj->bti", Lm, eps) * np.sqrt(dt)
    return out, means, raw, nxt


EM_KINDS = {"ou": 1, "lv": 2, "linear_diagonal": 3}


def euler_maruyama(kind: str, x0, theta, noise, dt: float, positive_dims=(), dtype=np.float32):
    """core/euler_maruyama.py:11-45 for the built-in model SDEs (``kind`` in EM_KINDS) -> trajectory [B, T+1, S]."""
    sfx, _ = _dt(dtype)
    x0 = _c(x0, dtype); theta = _c(theta, dtype); noise = _c(noise, dtype)
    B, T, S = noise.shape
    traj = np.empty((B, T + 1, S), dtype)
    fn = getattr(lib(), "vsde_oracle_em_fwd" + sfx); fn.restype = None
    fn(ctypes.c_int(EM_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S), ctypes.c_int(theta.shape[1]),
       _p(x0), _p(theta), _p(noise), ctypes.c_double(dt), _p(_mask(positive_dims, S)), _p(traj))
    return traj


def euler_maruyama_bwd(kind: str, theta, noise, traj, g_traj, dt: float, positive_dims=(), dtype=np.float32):
    """Reverse-mode gradient of ``euler_maruyama`` -> (g_x0 [B,S], g_theta [B,P])."""
    sfx, _ = _dt(dtype)
    theta = _c(theta, dtype); noise = _c(noise, dtype); traj = _c(traj, dtype); g_traj = _c(g_traj, dtype)
    B, T, S = noise.shape
    assert S <= 64
    g_x0 = np.empty((B, S), dtype); g_theta = np.empty_like(theta)
    fn = getattr(lib(), "vsde_oracle_em_bwd" + sfx); fn.restype = None
    fn(ctypes.c_int(EM_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S), ctypes.c_int(theta.shape[1]),
       _p(theta), _p(noise), _p(traj), _p(g_traj), ctypes.c_double(dt), _p(_mask(positive_dims, S)), _p(g_x0), _p(g_theta))
    return g_x0, g_theta


def sde_coefficients(kind: str, x, theta):
    """Drift [B,T,S] and diffusion factor [B,T,S,S] of the example SDEs on the first T points of x [B,T+1,S], float64:
    examples/ornstein_uhlenbeck.py:18-30 (f = kappa (mu - x), G = sigma), examples/lotka_volterra.py:18-46 (analytic 2x2
    Cholesky factor with three clamp(min=1e-6)); ``linear_diagonal`` is BASELINE config 5 (f = -a x, G = diag(softplus(b) + 1e-3)).
    This is what inference/evidence_lower_bound.py:37-40 evaluates on the flattened [(B T), S] states."""
    x = np.asarray(x, np.float64)[:, :-1]; th = np.asarray(theta, np.float64)[:, None, :]
    B, T, S = x.shape
    G = np.zeros((B, T, S, S))
    if kind == "ou":
        f = th[..., 0:1] * (th[..., 1:2] - x); G[..., 0, 0] = th[..., 2]
    elif kind == "lv":
        u, v = x[..., 0], x[..., 1]; t1, t2, t3 = th[..., 0], th[..., 1], th[..., 2]
        uv = t2 * u * v
        l00 = np.sqrt(np.maximum(t1 * u + uv, 1e-6)); l10 = -uv / np.maximum(l00, 1e-6)
        l11 = np.sqrt(np.maximum(t3 * v + uv - l10 * l10, 1e-6))
        f = np.stack([t1 * u - uv, uv - t3 * v], -1)
        G[..., 0, 0], G[..., 1, 0], G[..., 1, 1] = l00, l10, l11
    else:
        a, b = th[..., :S], th[..., S:]
        f = -a * x
        d = np.arange(S)
        G[..., d, d] = np.broadcast_to(np.logaddexp(0.0, b) + 1e-3, (B, T, S))
    return f, G


def sde_coefficients_bwd(kind: str, x, theta, g_drift, g_diffusion):
    """What torch autograd returns for ``sde_coefficients``: (g_x [B,T+1,S] with a zero last row, g_theta [B,P]); clamp(min)
    passes the gradient where its input is >= the bound (ATen clamp_backward)."""
    xf = np.asarray(x, np.float64); x = xf[:, :-1]; th = np.asarray(theta, np.float64)[:, None, :]
    gf = np.asarray(g_drift, np.float64); gG = np.asarray(g_diffusion, np.float64)
    B, T, S = x.shape
    gx = np.zeros_like(xf); gth = np.zeros((B, th.shape[-1]))
    if kind == "ou":
        gth[:, 0] = (gf[..., 0] * (th[..., 1] - x[..., 0])).sum(1); gth[:, 1] = (gf[..., 0] * th[..., 0]).sum(1)
        gth[:, 2] = gG[..., 0, 0].sum(1)
        gx[:, :-1, 0] = -gf[..., 0] * th[..., 0]
    elif kind == "lv":
        u, v = x[..., 0], x[..., 1]; t1, t2, t3 = th[..., 0], th[..., 1], th[..., 2]
        uv = t2 * u * v
        q00 = t1 * u + uv; l00 = np.sqrt(np.maximum(q00, 1e-6)); c = np.maximum(l00, 1e-6); l10 = -uv / c
        q11 = t3 * v + uv - l10 * l10; l11 = np.sqrt(np.maximum(q11, 1e-6))
        d_l00, d_l10, d_l11 = gG[..., 0, 0].copy(), gG[..., 1, 0].copy(), gG[..., 1, 1]
        d_q11 = np.where(q11 >= 1e-6, d_l11 / (2 * l11), 0.0)
        d_t3 = d_q11 * v; d_v = d_q11 * t3; d_uv = d_q11.copy(); d_l10 += -2 * l10 * d_q11
        d_uv += -d_l10 / c
        d_l00 += np.where(l00 >= 1e-6, d_l10 * uv / (c * c), 0.0)
        d_q00 = np.where(q00 >= 1e-6, d_l00 / (2 * l00), 0.0)
        d_t1 = d_q00 * u; d_u = d_q00 * t1; d_uv += d_q00
        d_t1 += gf[..., 0] * u; d_u += gf[..., 0] * t1; d_uv -= gf[..., 0]
        d_uv += gf[..., 1]; d_t3 -= gf[..., 1] * v; d_v -= gf[..., 1] * t3
        gth[:, 0], gth[:, 1], gth[:, 2] = d_t1.sum(1), (d_uv * u * v).sum(1), d_t3.sum(1)
        gx[:, :-1, 0] = d_u + d_uv * t2 * v; gx[:, :-1, 1] = d_v + d_uv * t2 * u
    else:
        a, b = th[..., :S], th[..., S:]
        d = np.arange(S)
        gth[:, :S] = -(gf * x).sum(1)
        gth[:, S:] = gG[..., d, d].sum(1) / (1.0 + np.exp(-b[:, 0]))
        gx[:, :-1] = -gf * a
    return gx, gth
