"""ctypes binding of libvsde_hip.so (see include/vsde_hip.h for the C ABI).

This is the only place the package touches native code.  There is deliberately NO CPU or
PyTorch fallback: if the library is missing, or tensors are not on a HIP device, the call
raises.  Tensors are passed as raw device pointers + the current torch stream handle.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

from .build import LIB_PATH

VSDE_ABI_VERSION = 1
MAX_LAYERS = 4      # reference: kernels/constants.py:13
MAX_HIDDEN = 1024
MAX_STATE = 32
DIAG_MIN = 1e-2     # reference: inference/constants.py:6


class HipLibraryError(RuntimeError):
    pass


class _Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "T", "S", "P", "C", "H", "L")]


class _Weights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0", "W_ih_stack", "W_hh_stack",
        "b_ih_stack", "b_hh_stack", "out_weight", "out_bias")]


class _CtxView(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("dtype", ctypes.c_int),
                ("batch_stride", ctypes.c_int64), ("step_stride", ctypes.c_int64)]


class _Grads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "x0", "context", "theta", "W_ih_l0", "W_hh_l0", "b_ih_l0", "b_hh_l0",
        "W_ih_stack", "W_hh_stack", "b_ih_stack", "b_hh_stack", "out_weight", "out_bias")] + [
        ("context_dtype", ctypes.c_int), ("context_batch_stride", ctypes.c_int64)]


EXPORTS = (
    "vsde_abi_version", "vsde_build_ablations", "vsde_last_error",
    "vsde_head_forward_workspace_bytes", "vsde_head_forward",
    "vsde_head_backward_workspace_bytes", "vsde_head_backward",
    "vsde_elbo_path_terms", "vsde_elbo_path_terms_bwd", "vsde_elbo_tail_fwd", "vsde_elbo_tail_bwd",
    "vsde_profile_enable", "vsde_profile_elapsed_ms", "vsde_debug_force_v1", "vsde_debug_head_mp", "vsde_head_mfma_range_exceeded",
    "vsde_ln_modulate_fwd", "vsde_ln_modulate_bwd", "vsde_gated_residual_fwd", "vsde_gated_residual_bwd",
    "vsde_swiglu_fwd", "vsde_swiglu_bwd", "vsde_gate_merge_fwd", "vsde_gate_merge_bwd",
    "vsde_qk_norm_rope_fwd", "vsde_qk_norm_rope_bwd_partials", "vsde_qk_norm_rope_bwd",
    "vsde_residual_ln_fwd", "vsde_residual_ln_bwd", "vsde_colsum_workspace_bytes", "vsde_linear_wgrad_workspace_bytes", "vsde_linear_wgrad_bf16", "vsde_linear_wgrad_bf16_rows", "vsde_linear_wgrad_group_workspace_bytes", "vsde_linear_wgrad_group_bf16",
    "vsde_attention_max_tokens", "vsde_attention_fwd_bf16", "vsde_attention_bwd_bf16",
    "vsde_attention_fused_supported", "vsde_attention_fwd_gated_bf16", "vsde_gate_bwd_delta", "vsde_attention_bwd_fused_partials",
    "vsde_attention_bwd_fused_bf16",
    "vsde_euler_maruyama_fwd", "vsde_euler_maruyama_bwd", "vsde_sde_coefficients_fwd", "vsde_sde_coefficients_bwd", "vsde_linear_bf16_supported", "vsde_linear_bf16", "vsde_linear_qknorm_bf16", "vsde_linear_gated_bf16", "vsde_linear_gate_bwd_bf16",
    "vsde_mlp_image_bytes", "vsde_mlp_fwd_bf16", "vsde_mlp_block_fwd_bf16", "vsde_mlp_attn_block_fwd_bf16", "vsde_linear_deep256_bf16", "vsde_mlp_debug_trace", "vsde_wgrad_debug_trace", "vsde_attn_debug_trace", "vsde_mlp_bwd_image_bytes", "vsde_mlp_bwd_bf16",
    "vsde_pack_tile_bytes", "vsde_pack_refresh", "vsde_optim_chunk_bytes", "vsde_optim_chunk_elems", "vsde_optim_step",
)

_lib: Optional[ctypes.CDLL] = None
_loaded_path: Optional[str] = None


def library_path() -> str:
    """The library file in use (the in-tree build unless VSDE_HIP_LIB names another one)."""
    return _loaded_path or os.environ.get("VSDE_HIP_LIB") or LIB_PATH


def has_ablations() -> bool:
    """Whether the loaded library is an ablation build (``python -m viforsdes_amd.build --ablations`` + VSDE_HIP_LIB): A/B switches are
    read from the environment and the losing kernel variants exist.  The shipped library says False."""
    return bool(load().vsde_build_ablations())


ABL_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libvsde_hip_abl.so")


def load() -> ctypes.CDLL:
    """Load the shared library (fails loudly if it was never built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VSDE_HIP_LIB") or LIB_PATH   # VSDE_HIP_LIB: an A/B build of the same sources (tools only)
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} is missing: build it with `python -m viforsdes_amd.build` "
            "(there is no CPU fallback for the fused head / ELBO kernels)")
    lib = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipLibraryError(f"{path} does not export {name}")
    global _loaded_path
    _loaded_path = path
    lib.vsde_abi_version.restype = ctypes.c_int
    if lib.vsde_abi_version() != VSDE_ABI_VERSION:
        raise HipLibraryError(f"{path}: ABI version mismatch; rebuild it")
    lib.vsde_last_error.restype = ctypes.c_char_p
    lib.vsde_head_forward_workspace_bytes.restype = ctypes.c_size_t
    lib.vsde_head_backward_workspace_bytes.restype = ctypes.c_size_t
    for f in (lib.vsde_head_forward, lib.vsde_head_backward, lib.vsde_elbo_path_terms, lib.vsde_elbo_path_terms_bwd):
        f.restype = ctypes.c_int
    lib.vsde_qk_norm_rope_bwd_partials.restype = ctypes.c_int64
    lib.vsde_attention_bwd_fused_partials.restype = ctypes.c_int64
    lib.vsde_linear_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.vsde_linear_wgrad_group_workspace_bytes.restype = ctypes.c_size_t
    lib.vsde_colsum_workspace_bytes.restype = ctypes.c_size_t
    lib.vsde_mlp_bwd_image_bytes.restype = ctypes.c_int64
    _lib = lib
    return lib


def _raise(rc: int) -> None:
    msg = load().vsde_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(msg)          # argument errors, like the reference's ValueError (models/head.py:33-36)
    raise HipLibraryError(f"HIP error {rc}: {msg}")


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(None if t is None or t.numel() == 0 else t.data_ptr())


def _require_hip(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipLibraryError(
                "the fused head / ELBO kernels only run on a HIP (cuda) device; got a "
                f"{t.device} tensor and there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("all tensors must live on the same device")
    assert dev is not None
    return dev


def _stream(dev: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def context_view(ctx: torch.Tensor):
    """Describe a [B, T, C] context (fp32 or bf16, last dim contiguous) without copying."""
    if ctx.dtype not in (torch.float32, torch.bfloat16):
        ctx = ctx.float()
    if ctx.stride(2) != 1:
        ctx = ctx.contiguous()
    v = _CtxView(ctypes.c_void_p(ctx.data_ptr()), 1 if ctx.dtype == torch.bfloat16 else 0,
                 ctx.stride(0), ctx.stride(1))
    return ctx, v


def _weights_struct(ws: Sequence[torch.Tensor]):
    keep = [_f32c(w) for w in ws]
    return keep, _Weights(*[_ptr(w) for w in keep])


def _dims(x0, ctx, theta, ws) -> _Dims:
    B, S = x0.shape
    T, C = ctx.shape[1], ctx.shape[2]
    P = theta.shape[1]
    H = ws[1].shape[1]
    L = 1 + (ws[4].shape[0] if ws[4].numel() > 0 else 0)
    if tuple(ws[0].shape) != (3 * H, S + C + P):
        raise ValueError(f"W_ih_l0 has shape {tuple(ws[0].shape)}, expected {(3 * H, S + C + P)}")
    return _Dims(B, T, S, P, C, H, L)


def head_forward(x0, ctx, theta, eps, ws, time_step: float, save: bool, diag_min: float = DIAG_MIN):
    """-> (paths[B,T+1,S], means[B,T,S], chol[B,T,S,S], chol_raw|None, acts|None), all fp32."""
    lib = load()
    dev = _require_hip(x0, ctx, theta, eps, *ws)
    x0 = _f32c(x0); theta = _f32c(theta); eps = _f32c(eps)
    ctx, cview = context_view(ctx)
    keep, wstruct = _weights_struct(ws)
    d = _dims(x0, ctx, theta, keep)
    B, T, S, H, L = d.B, d.T, d.S, d.H, d.L
    ntril = S * (S + 1) // 2
    with torch.cuda.device(dev):
        paths = torch.empty(B, T + 1, S, device=dev, dtype=torch.float32)
        means = torch.empty(B, T, S, device=dev, dtype=torch.float32)
        chol = torch.empty(B, T, S, S, device=dev, dtype=torch.float32)
        chol_raw = torch.empty(B, T, ntril, device=dev, dtype=torch.float32) if save else None
        acts = torch.empty(B, T, L, 5, H, device=dev, dtype=torch.float32) if save else None
        nbytes = lib.vsde_head_forward_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            _raise(-1)
        wsbuf = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        rc = lib.vsde_head_forward(
            ctypes.byref(d), _ptr(x0), ctypes.byref(cview), _ptr(theta), _ptr(eps), ctypes.byref(wstruct),
            ctypes.c_double(time_step), ctypes.c_double(diag_min), ctypes.c_int(1 if save else 0),
            _ptr(paths), _ptr(means), _ptr(chol), _ptr(chol_raw), _ptr(acts),
            _ptr(wsbuf), ctypes.c_size_t(nbytes), _stream(dev))
    if rc != 0:
        _raise(rc)
    return paths, means, chol, chol_raw, acts


def head_backward(g_paths, g_means, g_chol, ctx, theta, eps, paths, chol_raw, acts, ws,
                  time_step: float, diag_min: float = DIAG_MIN, context_grad_out=None):
    """-> 13 fp32 gradients in launch_bwd order (reference kernels/backward.py:766-784).

    ``context_grad_out``: optional contiguous [B, T+1, C] (f32 or bf16) tensor; the context gradient is then written
    straight into its first T steps (in its dtype) and returned in place of the fp32 [B,T,C] tensor; its last step is
    the caller's to zero."""
    lib = load()
    dev = _require_hip(g_paths, g_means, g_chol, ctx, theta, eps, paths, chol_raw, acts, *ws)
    g_paths = _f32c(g_paths); g_means = _f32c(g_means); g_chol = _f32c(g_chol)
    theta = _f32c(theta); eps = _f32c(eps)
    ctx, cview = context_view(ctx)
    keep, wstruct = _weights_struct(ws)
    B, T1, S = paths.shape
    d = _dims(paths[:, 0], ctx, theta, keep)
    T, C, P, H, L = d.T, d.C, d.P, d.H, d.L
    NO = S + S * (S + 1) // 2
    with torch.cuda.device(dev):
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        g = [mk(B, S), mk(B, T, C), mk(B, P), mk(3 * H, S + C + P), mk(3 * H, H), mk(3 * H), mk(3 * H),
             mk(L - 1, 3 * H, H), mk(L - 1, 3 * H, H), mk(L - 1, 3 * H), mk(L - 1, 3 * H), mk(NO, H), mk(NO)]
        cdt, cbs = 0, 0
        if context_grad_out is not None:
            o = context_grad_out
            if (not o.is_contiguous() or o.ndim != 3 or o.shape[0] != B or o.shape[1] < T or o.shape[2] != C
                    or o.dtype not in (torch.float32, torch.bfloat16) or o.device != dev):
                raise ValueError("context_grad_out must be a contiguous [B, >=T, C] f32/bf16 tensor on the same device")
            g[1] = o
            cdt, cbs = int(o.dtype == torch.bfloat16), o.shape[1] * C
        gstruct = _Grads(*[_ptr(t) for t in g], cdt, cbs)
        nbytes = lib.vsde_head_backward_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            _raise(-1)
        wsbuf = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        rc = lib.vsde_head_backward(
            ctypes.byref(d), _ptr(g_paths), _ptr(g_means), _ptr(g_chol), ctypes.byref(cview), _ptr(theta),
            _ptr(eps), _ptr(paths), _ptr(chol_raw), _ptr(acts), ctypes.byref(wstruct),
            ctypes.c_double(time_step), ctypes.c_double(diag_min), ctypes.byref(gstruct),
            _ptr(wsbuf), ctypes.c_size_t(nbytes), _stream(dev))
    if rc != 0:
        _raise(rc)
    return tuple(g)


def _mask_bytes(positive_dims, S: int):
    m = (ctypes.c_uint8 * S)()
    for dd in positive_dims or []:
        m[dd] = 1
    return m


def elbo_path_terms(z, x, means, chol, drift, diffusion, positive_dims, time_step: float):
    lib = load()
    dev = _require_hip(z, x, means, chol, drift, diffusion)
    z, x, means, chol, drift, diffusion = (_f32c(t) for t in (z, x, means, chol, drift, diffusion))
    B, T1, S = z.shape
    with torch.cuda.device(dev):
        outs = [torch.empty(B, device=dev, dtype=torch.float32) for _ in range(3)]
        rc = lib.vsde_elbo_path_terms(
            ctypes.c_int(B), ctypes.c_int(T1 - 1), ctypes.c_int(S), _ptr(z), _ptr(x), _ptr(means), _ptr(chol),
            _ptr(drift), _ptr(diffusion), _mask_bytes(positive_dims, S), ctypes.c_double(time_step),
            *[_ptr(o) for o in outs], _stream(dev))
    if rc != 0:
        _raise(rc)
    return tuple(outs)


def elbo_path_terms_bwd(z, x, means, chol, drift, diffusion, positive_dims, time_step: float, g_sde, g_gen, g_jac):
    lib = load()
    dev = _require_hip(z, x, means, chol, drift, diffusion, g_sde, g_gen, g_jac)
    z, x, means, chol, drift, diffusion, g_sde, g_gen, g_jac = (
        _f32c(t) for t in (z, x, means, chol, drift, diffusion, g_sde, g_gen, g_jac))
    B, T1, S = z.shape
    with torch.cuda.device(dev):
        outs = [torch.empty_like(z), torch.empty_like(x), torch.empty_like(means), torch.empty_like(chol),
                torch.empty_like(drift), torch.empty_like(diffusion)]
        rc = lib.vsde_elbo_path_terms_bwd(
            ctypes.c_int(B), ctypes.c_int(T1 - 1), ctypes.c_int(S), _ptr(z), _ptr(x), _ptr(means), _ptr(chol),
            _ptr(drift), _ptr(diffusion), _mask_bytes(positive_dims, S), ctypes.c_double(time_step),
            _ptr(g_sde), _ptr(g_gen), _ptr(g_jac), *[_ptr(o) for o in outs], _stream(dev))
    if rc != 0:
        _raise(rc)
    return tuple(outs)


SDE_KINDS = {"ornstein_uhlenbeck": 1, "lotka_volterra": 2, "linear_diagonal": 3}


def euler_maruyama_fwd(kind: str, x0, theta, noise, time_step: float, positive_dims=()):
    """Trajectory [B, T+1, S] of a built-in model SDE (``kind`` in SDE_KINDS) for given noise [B, T, S]."""
    lib = load()
    dev = _require_hip(x0, theta, noise)
    x0, theta, noise = _f32c(x0), _f32c(theta), _f32c(noise)
    B, T, S = noise.shape
    with torch.cuda.device(dev):
        traj = torch.empty(B, T + 1, S, device=dev, dtype=torch.float32)
        _call(lib.vsde_euler_maruyama_fwd, ctypes.c_int(SDE_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S),
              ctypes.c_int(theta.shape[1]), _ptr(x0), _ptr(theta), _ptr(noise), ctypes.c_double(time_step),
              _mask_bytes(positive_dims, S), _ptr(traj), _stream(dev))
    return traj


def euler_maruyama_bwd(kind: str, theta, noise, traj, g_traj, time_step: float, positive_dims=()):
    """(g_x0 [B,S], g_theta [B,P]) of ``euler_maruyama_fwd`` for the upstream gradient g_traj [B, T+1, S]."""
    lib = load()
    dev = _require_hip(theta, noise, traj, g_traj)
    theta, noise, traj, g_traj = _f32c(theta), _f32c(noise), _f32c(traj), _f32c(g_traj)
    B, T, S = noise.shape
    with torch.cuda.device(dev):
        g_x0 = torch.empty(B, S, device=dev, dtype=torch.float32)
        g_theta = torch.empty_like(theta)
        _call(lib.vsde_euler_maruyama_bwd, ctypes.c_int(SDE_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S),
              ctypes.c_int(theta.shape[1]), _ptr(theta), _ptr(noise), _ptr(traj), _ptr(g_traj), ctypes.c_double(time_step),
              _mask_bytes(positive_dims, S), _ptr(g_x0), _ptr(g_theta), _stream(dev))
    return g_x0, g_theta


def _tail_args(x_obs, obs_values, obs_matrix, variance, theta, prior_type, prior_mean, prior_std, post_mean, post_log_std, theta_positive_dims):
    B, K, S = x_obs.shape
    O, P = obs_values.shape[1], theta.shape[1]
    return (ctypes.c_int(B), ctypes.c_int(K), ctypes.c_int(S), ctypes.c_int(O), ctypes.c_int(P), _ptr(x_obs), _ptr(obs_values),
            _ptr(obs_matrix), ctypes.c_double(variance), _ptr(theta), ctypes.c_int(prior_type), ctypes.c_double(prior_mean),
            ctypes.c_double(prior_std), _ptr(post_mean), _ptr(post_log_std), _mask_bytes(theta_positive_dims, P))


def elbo_tail_fwd(x_obs, obs_values, obs_matrix, variance: float, theta, prior_type: int, prior_mean: float, prior_std: float,
                  post_mean, post_log_std, theta_positive_dims, sde_lp, gen_lp, log_jac):
    """[elbo, obs, sde, gen, prior, post] batch means (see include/vsde_hip.h: vsde_elbo_tail_fwd)."""
    lib = load()
    dev = _require_hip(x_obs, obs_values, theta, post_mean, post_log_std, sde_lp, gen_lp, log_jac)
    x_obs, obs_values, theta, post_mean, post_log_std, sde_lp, gen_lp, log_jac = (
        _f32c(t) for t in (x_obs, obs_values, theta, post_mean, post_log_std, sde_lp, gen_lp, log_jac))
    obs_matrix = None if obs_matrix is None else _f32c(obs_matrix)
    with torch.cuda.device(dev):
        out = torch.empty(6, device=dev, dtype=torch.float32)
        _call(lib.vsde_elbo_tail_fwd, *_tail_args(x_obs, obs_values, obs_matrix, variance, theta, prior_type, prior_mean, prior_std,
                                                  post_mean, post_log_std, theta_positive_dims),
              _ptr(sde_lp), _ptr(gen_lp), _ptr(log_jac), _ptr(out), _stream(dev))
    return out


def elbo_tail_bwd(x_obs, obs_values, obs_matrix, variance: float, theta, prior_type: int, prior_mean: float, prior_std: float,
                  post_mean, post_log_std, theta_positive_dims, g_out):
    """Gradients of ``elbo_tail_fwd``: (g_x_obs, g_theta, g_post_mean, g_post_log_std, g_sde, g_gen, g_jac)."""
    lib = load()
    dev = _require_hip(x_obs, obs_values, theta, post_mean, post_log_std, g_out)
    x_obs, obs_values, theta, post_mean, post_log_std, g_out = (
        _f32c(t) for t in (x_obs, obs_values, theta, post_mean, post_log_std, g_out))
    obs_matrix = None if obs_matrix is None else _f32c(obs_matrix)
    B = theta.shape[0]
    with torch.cuda.device(dev):
        g_x = torch.empty_like(x_obs); g_theta = torch.empty_like(theta)
        g_mean = torch.empty_like(post_mean); g_ls = torch.empty_like(post_log_std)
        g_paths = torch.empty(3, B, device=dev, dtype=torch.float32)
        _call(lib.vsde_elbo_tail_bwd, *_tail_args(x_obs, obs_values, obs_matrix, variance, theta, prior_type, prior_mean, prior_std,
                                                  post_mean, post_log_std, theta_positive_dims),
              _ptr(g_out), _ptr(g_x), _ptr(g_theta), _ptr(g_mean), _ptr(g_ls), _ptr(g_paths[0]), _ptr(g_paths[1]), _ptr(g_paths[2]),
              _stream(dev))
    return g_x, g_theta, g_mean, g_ls, g_paths[0], g_paths[1], g_paths[2]


def sde_coefficients_fwd(kind: str, x, theta):
    """(drift [B,T,S], diffusion [B,T,S,S]) of a built-in SDE on the first T points of every path x [B, T+1, S]."""
    lib = load()
    dev = _require_hip(x, theta)
    x, theta = _f32c(x), _f32c(theta)
    B, T1, S = x.shape
    T = T1 - 1
    with torch.cuda.device(dev):
        drift = torch.empty(B, T, S, device=dev, dtype=torch.float32)
        diffusion = torch.empty(B, T, S, S, device=dev, dtype=torch.float32)
        _call(lib.vsde_sde_coefficients_fwd, ctypes.c_int(SDE_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T), ctypes.c_int(S),
              ctypes.c_int(theta.shape[1]), _ptr(x), _ptr(theta), _ptr(drift), _ptr(diffusion), _stream(dev))
    return drift, diffusion


def sde_coefficients_bwd(kind: str, x, theta, g_drift, g_diffusion):
    """(g_x [B,T+1,S], g_theta [B,P]) of ``sde_coefficients_fwd``."""
    lib = load()
    dev = _require_hip(x, theta, g_drift, g_diffusion)
    x, theta, g_drift, g_diffusion = _f32c(x), _f32c(theta), _f32c(g_drift), _f32c(g_diffusion)
    B, T1, S = x.shape
    with torch.cuda.device(dev):
        g_x = torch.empty_like(x)
        g_theta = torch.empty_like(theta)
        _call(lib.vsde_sde_coefficients_bwd, ctypes.c_int(SDE_KINDS[kind]), ctypes.c_int(B), ctypes.c_int(T1 - 1), ctypes.c_int(S),
              ctypes.c_int(theta.shape[1]), _ptr(x), _ptr(theta), _ptr(g_drift), _ptr(g_diffusion), _ptr(g_x), _ptr(g_theta),
              _stream(dev))
    return g_x, g_theta


def pack_refresh(table: torch.Tensor) -> None:
    """One launch that re-fills the cached bf16 GEMM operands (and their transposes) from the fp32 parameters; ``table`` is
    the int64 [n_tiles, 8] device tensor built by ``primitives.fused.PackedWeight.refresh_all``."""
    lib = load()
    dev = _require_hip(table)
    if table.dtype != torch.int64 or table.ndim != 2 or table.shape[1] * 8 != lib.vsde_pack_tile_bytes() or not table.is_contiguous():
        raise ValueError("pack-refresh table must be a contiguous int64 [n_tiles, 8] tensor")
    with torch.cuda.device(dev):
        _call(lib.vsde_pack_refresh, _ptr(table), ctypes.c_int(table.shape[0]), _stream(dev))


def optim_chunk_elems() -> int:
    return int(load().vsde_optim_chunk_elems())


def optim_step(table: torch.Tensor, grad_ptrs: torch.Tensor, scale: Optional[torch.Tensor], partials: torch.Tensor, tstate: torch.Tensor,
               groups: torch.Tensor, max_norm: float, ema_weight: float, out: torch.Tensor) -> None:
    """unscale + global-norm clip + AdamW + EMA of every parameter as two launches (csrc/vsde_optim.hip).  ``table``: int64
    [n_chunks, 8] chunk records; ``grad_ptrs``: int64 [n_params] device pointers of this step's gradients; ``scale``: the loss
    scale (fp32 device scalar) or None; ``groups``: float64 [n_groups, 5]; ``out``: fp32 [2] <- (gradient norm, found_inf)."""
    lib = load()
    dev = _require_hip(table, grad_ptrs, partials, tstate, groups, out)
    if (table.dtype != torch.int64 or table.ndim != 2 or table.shape[1] * 8 != lib.vsde_optim_chunk_bytes() or not table.is_contiguous()
            or grad_ptrs.dtype != torch.int64 or groups.dtype != torch.float64 or partials.numel() < table.shape[0]
            or tstate.dtype != torch.float32 or out.dtype != torch.float32):
        raise ValueError("bad optimizer-step buffers")
    with torch.cuda.device(dev):
        _call(lib.vsde_optim_step, _ptr(table), ctypes.c_int(table.shape[0]), _ptr(grad_ptrs), _ptr(scale), _ptr(partials), _ptr(tstate),
              _ptr(groups), ctypes.c_double(max_norm), ctypes.c_double(ema_weight), _ptr(out), _stream(dev))


def profile_enable(on: bool) -> None:
    load().vsde_profile_enable(ctypes.c_int(1 if on else 0))


def profile_elapsed_ms(which: int) -> float:
    """Duration of the last timed serial kernel (0 = forward/train, 1 = backward), HIP events."""
    ms = ctypes.c_float(0.0)
    rc = load().vsde_profile_elapsed_ms(ctypes.c_int(which), ctypes.byref(ms))
    if rc != 0:
        _raise(rc)
    return float(ms.value)


def debug_force_v1(on: bool) -> None:
    """Test hook: run 1-2 layer heads through the kernels that normally serve 3-4 layers."""
    load().vsde_debug_force_v1(ctypes.c_int(1 if on else 0))


def debug_head_mp(mode: int) -> None:
    """Forward time-stepping kernel for hidden_dim 64 / L <= 2 / state_dim <= 2: 1 = the multi-path MFMA kernel whenever
    applicable, 0 = the four-waves-per-path kernel, -1 = default (VSDE_HEAD_MP, else by batch size)."""
    load().vsde_debug_head_mp(ctypes.c_int(mode))


def head_mfma_range_exceeded(clear: bool = False) -> bool:
    """True once a GRU head weight has left the f16 range of the multi-path MFMA kernels (|W| > 2.2e4): the launch that found it returned
    non-finite paths and every later launch takes the fp32 kernels.  Host-mapped flag written in stream order, no synchronisation: callers
    that replay captured graphs poll it after a replay.  ``clear=True`` resets it (tests)."""
    return bool(load().vsde_head_mfma_range_exceeded(ctypes.c_int(1 if clear else 0)))


# ---------------------------------------------------------------------------------------------
# fused encoder operators (csrc/vsde_encoder.hip)
def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise ValueError(f"fused encoder ops support float32/bfloat16, got {t.dtype}")


def _call(fn, *args):
    rc = fn(*args)
    if rc != 0:
        _raise(rc)


def _i64(v):
    return ctypes.c_int64(int(v))


def _mod_pitch(C, *vecs):
    """Common row pitch of per-batch-row vectors [B, C] (contiguous, or column ranges of one [B, pitch] buffer)."""
    pitch = None
    for v in vecs:
        if v is None:
            continue
        if v.ndim != 2 or v.shape[1] != C or v.stride(1) != 1 or v.stride(0) < C:
            raise ValueError(f"expected a [B,{C}] row-pitched vector, got shape {tuple(v.shape)} strides {v.stride()}")
        if pitch is None:
            pitch = v.stride(0)
        elif v.stride(0) != pitch:
            raise ValueError("scale / shift / gate and their gradient outputs must share one row pitch")
    return pitch


def _like_vec(v):
    return torch.empty_strided(v.shape, v.stride(), device=v.device, dtype=v.dtype)


def ln_modulate_fwd(x, scale, shift, eps):
    lib = load(); dev = _require_hip(x, scale, shift)
    B, N, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B, N, device=dev, dtype=torch.float32); rstd = torch.empty_like(mean)
    with torch.cuda.device(dev):
        _call(lib.vsde_ln_modulate_fwd, _dt(x), _ptr(x), _ptr(scale), _ptr(shift), _ptr(y), _ptr(mean), _ptr(rstd), _i64(B),
              ctypes.c_int(N), ctypes.c_int(C), ctypes.c_double(eps), _i64(_mod_pitch(C, scale, shift)), _stream(dev))
    return y, mean, rstd


def _colsum_workspace(lib, B, C, dev):
    nbytes = lib.vsde_colsum_workspace_bytes(_i64(B), ctypes.c_int(C))
    return torch.empty(max(int(nbytes), 1), device=dev, dtype=torch.uint8)


def ln_modulate_bwd(x, scale, dy, mean, rstd, dres=None, dscale=None, dshift=None):
    """dres (optional, same shape as x) is added to dx inside the kernel; dscale / dshift: optional destinations with the row
    pitch of ``scale`` (column ranges of a shared gradient buffer)."""
    lib = load(); dev = _require_hip(x, scale, dy)
    B, N, C = x.shape
    dx = torch.empty_like(x)
    dscale = _like_vec(scale) if dscale is None else dscale
    dshift = _like_vec(scale) if dshift is None else dshift
    ws = _colsum_workspace(lib, B, C, dev)
    with torch.cuda.device(dev):
        _call(lib.vsde_ln_modulate_bwd, _dt(x), _ptr(x), _ptr(scale), _ptr(dy), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx),
              _ptr(dscale), _ptr(dshift), _i64(B), ctypes.c_int(N), ctypes.c_int(C), _i64(_mod_pitch(C, scale, dscale, dshift)),
              _ptr(ws), ctypes.c_size_t(ws.numel()), _stream(dev))
    return dx, dscale, dshift


def residual_ln_fwd(x, y, gate, scale, shift, eps):
    """(xnew, h, mean, rstd) with xnew = x + gate*y and h = LN(xnew)*(1+scale)+shift."""
    lib = load(); dev = _require_hip(x, y, gate, scale, shift)
    B, N, C = x.shape
    xnew = torch.empty_like(x); h = torch.empty_like(x)
    mean = torch.empty(B * N, device=dev, dtype=torch.float32); rstd = torch.empty_like(mean)
    with torch.cuda.device(dev):
        _call(lib.vsde_residual_ln_fwd, _dt(x), _ptr(x), _ptr(y), _ptr(gate), _ptr(scale), _ptr(shift), _ptr(xnew), _ptr(h),
              _ptr(mean), _ptr(rstd), _i64(B), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_double(eps),
              _i64(_mod_pitch(C, gate, scale, shift)), _stream(dev))
    return xnew, h, mean, rstd


def residual_ln_bwd(xnew, y, gate, scale, dh, dxnew, mean, rstd, dgate=None, dscale=None, dshift=None):
    """(dx, dy, dgate, dscale, dshift); dxnew may be None; optional preallocated d-vector destinations (pitch of ``gate``)."""
    lib = load(); dev = _require_hip(xnew, y, gate, scale, dh)
    B, N, C = xnew.shape
    dx = torch.empty_like(xnew); dy = torch.empty_like(xnew)
    dgate = _like_vec(gate) if dgate is None else dgate
    dscale = _like_vec(scale) if dscale is None else dscale
    dshift = _like_vec(scale) if dshift is None else dshift
    ws = _colsum_workspace(lib, B, C, dev)
    with torch.cuda.device(dev):
        _call(lib.vsde_residual_ln_bwd, _dt(xnew), _ptr(xnew), _ptr(y), _ptr(gate), _ptr(scale), _ptr(dh), _ptr(dxnew),
              _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dy), _ptr(dgate), _ptr(dscale), _ptr(dshift), _i64(B), ctypes.c_int(N),
              ctypes.c_int(C), _i64(_mod_pitch(C, gate, scale, dgate, dscale, dshift)), _ptr(ws), ctypes.c_size_t(ws.numel()),
              _stream(dev))
    return dx, dy, dgate, dscale, dshift


def gated_residual_fwd(x, y, gate):
    lib = load(); dev = _require_hip(x, y, gate)
    B, N, C = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        _call(lib.vsde_gated_residual_fwd, _dt(x), _ptr(x), _ptr(y), _ptr(gate), _ptr(out), _i64(B), ctypes.c_int(N),
              ctypes.c_int(C), _i64(_mod_pitch(C, gate)), _stream(dev))
    return out


def gated_residual_bwd(y, gate, dout, dgate=None):
    lib = load(); dev = _require_hip(y, gate, dout)
    B, N, C = y.shape
    dy = torch.empty_like(y)
    dgate = _like_vec(gate) if dgate is None else dgate
    ws = _colsum_workspace(lib, B, C, dev)
    with torch.cuda.device(dev):
        _call(lib.vsde_gated_residual_bwd, _dt(y), _ptr(y), _ptr(gate), _ptr(dout), _ptr(dy), _ptr(dgate), _i64(B),
              ctypes.c_int(N), ctypes.c_int(C), _i64(_mod_pitch(C, gate, dgate)), _ptr(ws), ctypes.c_size_t(ws.numel()),
              _stream(dev))
    return dy, dgate


def swiglu_fwd(u):
    lib = load(); dev = _require_hip(u)
    H2 = u.shape[-1] // 2
    out = torch.empty(*u.shape[:-1], H2, device=dev, dtype=u.dtype)
    with torch.cuda.device(dev):
        _call(lib.vsde_swiglu_fwd, _dt(u), _ptr(u), _ptr(out), _i64(u.numel() // (2 * H2)), ctypes.c_int(H2), _stream(dev))
    return out


def swiglu_bwd(u, dout):
    lib = load(); dev = _require_hip(u, dout)
    H2 = u.shape[-1] // 2
    du = torch.empty_like(u)
    with torch.cuda.device(dev):
        _call(lib.vsde_swiglu_bwd, _dt(u), _ptr(u), _ptr(dout), _ptr(du), _i64(u.numel() // (2 * H2)), ctypes.c_int(H2), _stream(dev))
    return du


def _head_dims(t, token_major):
    """(B, heads, N, d) of a per-head tensor stored [B,heads,N,d] (token_major False) or [B,N,heads,d] (True)."""
    if token_major:
        B, N, h, d = t.shape
    else:
        B, h, N, d = t.shape
    return B, h, N, d


def _row_pitch(t, width):
    """Row pitch (elements) of a [B,N,width] tensor that is either contiguous or a column range of a contiguous
    [B,N,W] buffer (the merged [qkv | gate] projection)."""
    B, N, w = t.shape
    pitch = t.stride(1)
    if w != width or t.stride(2) != 1 or pitch < width or t.stride(0) != N * pitch:
        raise ValueError(f"expected a [B,N,{width}] row-pitched tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    return pitch


def gate_merge_fwd(attn, glog, token_major=False):
    lib = load(); dev = _require_hip(attn, glog)
    B, h, N, d = _head_dims(attn, token_major)
    out = torch.empty(B, N, h * d, device=dev, dtype=attn.dtype)
    with torch.cuda.device(dev):
        _call(lib.vsde_gate_merge_fwd, _dt(attn), _ptr(attn), _ptr(glog), _ptr(out), _i64(B), ctypes.c_int(N), ctypes.c_int(h),
              ctypes.c_int(d), ctypes.c_int(int(token_major)), _i64(_row_pitch(glog, d)), _stream(dev))
    return out


def gate_merge_bwd(attn, glog, dout, token_major=False, dglog=None):
    """dglog: optional preallocated destination with the same row pitch as glog (a column range of a shared buffer)."""
    lib = load(); dev = _require_hip(attn, glog, dout)
    B, h, N, d = _head_dims(attn, token_major)
    dattn = torch.empty_like(attn)
    if dglog is None:
        dglog = torch.empty_like(glog) if glog.is_contiguous() else torch.empty_strided(glog.shape, glog.stride(), device=dev, dtype=glog.dtype)
    pitch = _row_pitch(glog, d)
    if _row_pitch(dglog, d) != pitch:
        raise ValueError("dglog must have the row pitch of glog")
    with torch.cuda.device(dev):
        _call(lib.vsde_gate_merge_bwd, _dt(attn), _ptr(attn), _ptr(glog), _ptr(dout), _ptr(dattn), _ptr(dglog), _i64(B),
              ctypes.c_int(N), ctypes.c_int(h), ctypes.c_int(d), ctypes.c_int(int(token_major)), _i64(pitch), _stream(dev))
    return dattn, dglog


def qk_norm_rope_fwd(qkv, cos, sin, wq, wk, v0, lam, heads, eps, token_major=False):
    lib = load(); dev = _require_hip(qkv, cos, sin, wq, wk)
    B, N, C3 = qkv.shape
    d = C3 // 3 // heads
    pitch = _row_pitch(qkv, C3)
    shape = (B, N, heads, d) if token_major else (B, heads, N, d)
    q = torch.empty(shape, device=dev, dtype=qkv.dtype); k = torch.empty_like(q); v = torch.empty_like(q)
    with torch.cuda.device(dev):
        _call(lib.vsde_qk_norm_rope_fwd, _dt(qkv), _ptr(qkv), _ptr(cos), _ptr(sin), _ptr(wq), _ptr(wk), _ptr(v0), _ptr(lam),
              _ptr(q), _ptr(k), _ptr(v), _i64(B), ctypes.c_int(N), ctypes.c_int(heads), ctypes.c_int(d), ctypes.c_double(eps),
              ctypes.c_int(int(token_major)), _i64(pitch), _stream(dev))
    return q, k, v


def qk_norm_rope_bwd(qkv, cos, sin, wq, wk, v0, lam, dq, dk, dv, heads, eps, token_major=False, dqkv=None, dv0=None,
                     dv_extra=None):
    """dqkv: optional preallocated destination with the same row pitch as qkv (a column range of a shared buffer);
    dv0: optional existing buffer to ACCUMULATE the value-residual gradient into; dv_extra: added to dv first."""
    lib = load(); dev = _require_hip(qkv, dq, dk, dv)
    B, N, C3 = qkv.shape
    d = C3 // 3 // heads
    pitch = _row_pitch(qkv, C3)
    if dqkv is None:
        dqkv = torch.empty_like(qkv) if qkv.is_contiguous() else torch.empty_strided(qkv.shape, qkv.stride(), device=dev, dtype=qkv.dtype)
    if _row_pitch(dqkv, C3) != pitch:
        raise ValueError("dqkv must have the row pitch of qkv")
    accumulate = dv0 is not None
    if v0 is not None and dv0 is None:
        dv0 = torch.empty_like(dv)
    nparts = lib.vsde_qk_norm_rope_bwd_partials(_i64(B), ctypes.c_int(N), ctypes.c_int(heads), ctypes.c_int(d))
    parts = torch.empty(nparts, device=dev, dtype=torch.float32) if v0 is not None else None
    with torch.cuda.device(dev):
        _call(lib.vsde_qk_norm_rope_bwd, _dt(qkv), _ptr(qkv), _ptr(cos), _ptr(sin), _ptr(wq), _ptr(wk), _ptr(v0), _ptr(lam),
              _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dqkv), _ptr(dv0), _ptr(parts), _i64(B), ctypes.c_int(N), ctypes.c_int(heads),
              ctypes.c_int(d), ctypes.c_double(eps), ctypes.c_int(int(token_major)), _i64(pitch), ctypes.c_int(int(accumulate)),
              _ptr(dv_extra), _stream(dev))
    dlam = parts.sum() if parts is not None else None
    return dqkv, dv0, dlam


def attention_max_tokens() -> int:
    return int(load().vsde_attention_max_tokens())


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float):
    """softmax(scale q k^T) v for token-major bf16 heads [B,N,H,d], d in (64, 128); returns (o, lse [B,H,N] fp32)."""
    lib = load(); dev = _require_hip(q, k, v)
    B, N, H, d = q.shape
    if q.dtype != torch.bfloat16 or not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
        raise ValueError("attention_fwd needs contiguous bf16 [B,N,H,d] tensors")
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_attention_fwd_bf16, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _i64(B), ctypes.c_int(N),
              ctypes.c_int(H), ctypes.c_int(d), ctypes.c_double(scale), _stream(dev))
    return o, lse


def attention_bwd(dout, q, k, v, o, lse, scale: float):
    """(dq, dk, dv) token-major bf16 for ``attention_fwd``'s inputs/outputs."""
    lib = load(); dev = _require_hip(dout, q, k, v, o, lse)
    B, N, H, d = q.shape
    for t in (dout, q, k, v, o):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or t.shape != q.shape:
            raise ValueError("attention_bwd needs contiguous bf16 [B,N,H,d] tensors of one shape")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_attention_bwd_bf16, _ptr(dout), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(dq), _ptr(dk),
              _ptr(dv), _ptr(delta), _i64(B), ctypes.c_int(N), ctypes.c_int(H), ctypes.c_int(d), ctypes.c_double(scale),
              _stream(dev))
    return dq, dk, dv


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool, row_map: Optional[torch.Tensor] = None,
                 out_rows: Optional[int] = None):
    """dW[N,K] = dy^T x and db[N] = colsum(dy) in fp32 for bf16 dy [M,N], x [M,K].  ``row_map`` (int32 [N] on the device):
    product row n is stored at output row ``row_map[n]`` (negative: dropped) of an ``[out_rows, K]`` result."""
    lib = load(); dev = _require_hip(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    rows = N if row_map is None else int(out_rows)
    dW = torch.empty(rows, K, device=dev, dtype=torch.float32)
    db = torch.empty(rows, device=dev, dtype=torch.float32) if want_bias else None
    nbytes = lib.vsde_linear_wgrad_workspace_bytes(_i64(M), ctypes.c_int(N), ctypes.c_int(K))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_wgrad_bf16_rows, _ptr(dy), _ptr(x), _i64(M), ctypes.c_int(N), ctypes.c_int(K), _ptr(dW), _ptr(db),
              _ptr(row_map), _ptr(ws), ctypes.c_size_t(nbytes), _stream(dev))
    return dW, db


class _WgradItem(ctypes.Structure):
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("M", ctypes.c_int64), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("row_map", ctypes.c_void_p)]


def linear_wgrad_group(problems, group_plan: bool = False):
    """``[(dy, x, want_bias, row_map, out_rows), ...]`` -> ``[(dW, db), ...]``: every problem as ``linear_wgrad`` computes it, all of
    them in a handful of launches (csrc/vsde_wgrad.hip, ``vsde_linear_wgrad_group_bf16``).  ``group_plan=False``: bit-identical to
    one call per problem; ``True``: split counts chosen for the group as a whole (other summation order, less partial traffic)."""
    lib = load()
    dev = _require_hip(*[t for pr in problems for t in pr[:2]])
    n = len(problems)
    items = (_WgradItem * n)()
    outs = []
    for i, (dy, x, want_bias, row_map, out_rows) in enumerate(problems):
        M, N = dy.shape
        K = x.shape[1]
        rows = N if row_map is None else int(out_rows)
        dW = torch.empty(rows, K, device=dev, dtype=torch.float32)
        db = torch.empty(rows, device=dev, dtype=torch.float32) if want_bias else None
        items[i] = _WgradItem(dy.data_ptr(), x.data_ptr(), M, N, K, dW.data_ptr(), None if db is None else db.data_ptr(),
                              None if row_map is None else row_map.data_ptr())
        outs.append((dW, db))
    nbytes = lib.vsde_linear_wgrad_group_workspace_bytes(ctypes.c_int(n), ctypes.byref(items))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_wgrad_group_bf16, ctypes.c_int(n), ctypes.byref(items), ctypes.c_int(1 if group_plan else 0), _ptr(ws),
              ctypes.c_size_t(nbytes), _stream(dev))
    return outs


EPI_PLAIN, EPI_SWIGLU, EPI_SWIGLU_BWD = 0, 1, 2


def linear_variant(M: int, N: int, K: int, epilogue: int = EPI_PLAIN) -> int:
    """0 = shape not covered, 1 = rows kernel (K in {128, 256, 512}), 2 = cols kernel, 3 = persistent deep-reduction kernel
    (K >= 512, N % 256 == 0, N <= K, M >= 32768)."""
    return int(load().vsde_linear_bf16_supported(_i64(M), ctypes.c_int(N), ctypes.c_int(K), ctypes.c_int(epilogue)))


def linear_supported(M: int, N: int, K: int, epilogue: int = EPI_PLAIN) -> bool:
    return linear_variant(M, N, K, epilogue) != 0


def _rows2d(t: torch.Tensor):
    """(tensor, row pitch) of a bf16 matrix whose rows are contiguous (a column range of a wider buffer is fine)."""
    if t.dtype != torch.bfloat16 or t.ndim != 2 or t.stride(1) != 1:
        raise ValueError(f"expected a row-pitched bf16 matrix, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t, t.stride(0)


def linear_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None):
    """y = x w^T + bias on the MFMA kernels: x [M,K] (row-pitched), w [N,K] contiguous, bias [N] or None -> y [M,N] bf16."""
    lib = load(); dev = _require_hip(x, w)
    x, ldx = _rows2d(x)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if out is None else out
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_bf16, _ptr(x), _i64(ldx), _ptr(w), _ptr(bias), _ptr(y), _i64(y.stride(0)), _i64(M), ctypes.c_int(N),
              ctypes.c_int(K), ctypes.c_int(EPI_PLAIN), None, _i64(0), None, _i64(0), _stream(dev))
    return y


def linear_swiglu_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], want_u: bool = True):
    """(u [M,N] or None, s [M,N/2]) for the interleaved-packed SwiGLU input projection w [N,K]."""
    lib = load(); dev = _require_hip(x, w)
    x, ldx = _rows2d(x)
    M, K = x.shape
    N = w.shape[0]
    u = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if want_u else None
    s = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_bf16, _ptr(x), _i64(ldx), _ptr(w), _ptr(bias), _ptr(u), _i64(N), _i64(M), ctypes.c_int(N),
              ctypes.c_int(K), ctypes.c_int(EPI_SWIGLU), _ptr(s), _i64(N // 2), None, _i64(0), _stream(dev))
    return u, s


def linear_qknorm_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], heads: int, tokens: int, cos, sin, wq, wk,
                       v0: Optional[torch.Tensor], lam: Optional[torch.Tensor], eps: float, save: bool = False):
    """Attention projection: x [M,256] against the packed [q | k | v | gate] weight w [3*heads*64 + G, 256] with the
    QK-RMS-norm + RoPE + value mix in the GEMM epilogue -> (q, k, v [M, heads*64] token-major, gate logits [M, G] or None).
    ``save`` (training): the gate block is returned as sigmoid(logits) and (rinv [M, 2*heads] fp32, vdiff [M, heads*64] bf16 or
    None) are returned for the fused backward."""
    lib = load(); dev = _require_hip(x, w, cos, sin, wq, wk)
    x, ldx = _rows2d(x)
    M, K = x.shape
    C = heads * 64
    G = w.shape[0] - 3 * C
    q = torch.empty(M, C, device=dev, dtype=torch.bfloat16); k = torch.empty_like(q); v = torch.empty_like(q)
    gate = torch.empty(M, G, device=dev, dtype=torch.bfloat16) if G > 0 else None
    rinv = torch.empty(M, 2 * heads, device=dev, dtype=torch.float32) if save else None
    vdiff = torch.empty(M, C, device=dev, dtype=torch.bfloat16) if (save and v0 is not None) else None
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_qknorm_bf16, _ptr(x), _i64(ldx), _ptr(w), _ptr(bias), _i64(M), ctypes.c_int(K), ctypes.c_int(heads),
              ctypes.c_int(G), ctypes.c_int(tokens), _ptr(cos), _ptr(sin), _ptr(wq), _ptr(wk), _ptr(v0), _ptr(lam),
              ctypes.c_double(eps), _ptr(q), _ptr(k), _ptr(v), _ptr(gate), _i64(G), ctypes.c_int(int(save)), _ptr(rinv), _ptr(vdiff),
              _stream(dev))
    if save:
        return q, k, v, gate, rinv, vdiff
    return q, k, v, gate


def attention_fused_supported(N: int, head_dim: int) -> bool:
    return bool(load().vsde_attention_fused_supported(ctypes.c_int(N), ctypes.c_int(head_dim)))


def attention_fwd_gated(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, gate: torch.Tensor, scale: float):
    """softmax(scale q k^T) v * gate for token-major bf16 heads [B,N,H,64] and the gate factors s = sigmoid(logits) [B*N, >=64]
    (row-pitched; ``linear_qknorm_bf16(save=True)`` produces them): returns (the merged rows og [B,N,H,64], lse [B,H,N] fp32)."""
    lib = load(); dev = _require_hip(q, k, v, gate)
    B, N, H, d = q.shape
    if q.dtype != torch.bfloat16 or d != 64 or not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
        raise ValueError("attention_fwd_gated needs contiguous bf16 [B,N,H,64] tensors")
    gate, ldg = _rows2d(gate)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_attention_fwd_gated_bf16, _ptr(q), _ptr(k), _ptr(v), _ptr(gate), _i64(ldg), _ptr(o), _ptr(lse), _i64(B),
              ctypes.c_int(N), ctypes.c_int(H), ctypes.c_double(scale), _stream(dev))
    return o, lse


def gate_bwd_delta(dout: torch.Tensor, og: torch.Tensor, gate: torch.Tensor, dgate: torch.Tensor):
    """Backward of the gate folded into ``attention_fwd_gated``: dout, og [B,N,H,64] bf16, gate factors s [B*N, >=64]; ``dgate``
    (row-pitched [B*N, 64], e.g. the gate columns of the projection's gradient buffer) receives the gradient of the LOGITS.
    Returns (dattn [B,N,H,64] = the gradient of the ungated attention output, delta [B,H,N] fp32 = <dattn, o>)."""
    lib = load(); dev = _require_hip(dout, og, gate, dgate)
    B, N, H, d = og.shape
    if d != 64 or not (dout.is_contiguous() and og.is_contiguous()) or dout.dtype != torch.bfloat16:
        raise ValueError("gate_bwd_delta needs contiguous bf16 [B,N,H,64] tensors")
    gate, ldg = _rows2d(gate); dgate, ldd = _rows2d(dgate)
    dattn = torch.empty_like(og)
    delta = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_gate_bwd_delta, _ptr(dout), _ptr(og), _ptr(gate), _i64(ldg), _ptr(dattn), _ptr(dgate), _i64(ldd), _ptr(delta),
              _i64(B), ctypes.c_int(N), ctypes.c_int(H), _stream(dev))
    return dattn, delta


def mlp_image_bytes(C: int) -> tuple[int, int, int]:
    """Bytes per 16-unit tile of the fused SwiGLU MLP's weight images (W1, W2, b1) for width C."""
    a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    rc = load().vsde_mlp_image_bytes(ctypes.c_int(C), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    if rc != 0:
        _raise(rc)
    return int(a.value), int(b.value), int(c.value)


def mlp_fwd(x: torch.Tensor, w1_img: torch.Tensor, w2_img: torch.Tensor, b1_img: torch.Tensor, b2: Optional[torch.Tensor], H: int,
            want_s: bool = False):
    """Fused SwiGLU MLP forward (csrc/vsde_mlp.hip): x [M,C] bf16 (row-pitched) -> (y [M,C] bf16, s [M,H] bf16 or None)."""
    lib = load(); dev = _require_hip(x, w1_img, w2_img, b1_img)
    x, ldx = _rows2d(x)
    M, C = x.shape
    y = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    s = torch.empty(M, H, device=dev, dtype=torch.bfloat16) if want_s else None
    with torch.cuda.device(dev):
        _call(lib.vsde_mlp_fwd_bf16, _ptr(x), _i64(ldx), _ptr(w1_img), _ptr(w2_img), _ptr(b1_img), _ptr(b2), _ptr(y), _i64(C), _ptr(s),
              _i64(H), _i64(M), ctypes.c_int(C), ctypes.c_int(H), _stream(dev))
    return y, s


def mlp_block_fwd(x, yin, ga, sc, sh, gm, sn, hs, eps: float, eps_next: float, w1_img, w2_img, b1_img, b2, H: int):
    """Block form of the fused SwiGLU MLP (no-grad): x, yin [B,N,C] bf16 contiguous, modulation vectors [B,C] views of one
    [B,pitch] buffer -> (tok [B,N,C], hnext [B,N,C] or None when sn / hs are None).  See include/vsde_hip.h."""
    lib = load(); dev = _require_hip(x, yin, ga, sc, sh, gm, w1_img, w2_img, b1_img)
    B, N, C = x.shape
    if not (x.is_contiguous() and yin.is_contiguous() and x.dtype == torch.bfloat16 and yin.dtype == torch.bfloat16 and yin.shape == x.shape):
        raise ValueError("mlp_block_fwd: x and yin must be contiguous bf16 tensors of one shape")
    mp = _mod_pitch(C, ga, sc, sh, gm, sn, hs)
    tok = torch.empty_like(x)
    hnext = torch.empty_like(x) if sn is not None else None
    with torch.cuda.device(dev):
        _call(lib.vsde_mlp_block_fwd_bf16, _ptr(x), _ptr(yin), _ptr(ga), _ptr(sc), _ptr(sh), _ptr(gm), _ptr(sn), _ptr(hs), _i64(mp),
              ctypes.c_int(N), ctypes.c_double(eps), ctypes.c_double(eps_next), _ptr(w1_img), _ptr(w2_img), _ptr(b1_img), _ptr(b2),
              _ptr(tok), _ptr(hnext), _i64(B * N), ctypes.c_int(C), ctypes.c_int(H), _stream(dev))
    return tok, hnext


def mlp_attn_block_fwd(x, attn, glog, wo_img, bo, ga, sc, sh, gm, sn, hs, eps: float, eps_next: float, w1_img, w2_img, b1_img, b2, H: int):
    """``mlp_block_fwd`` with the attention out projection in front: attn [B,N,C] bf16 contiguous (merged heads), glog [B*N,64]
    gate logits (row pitch free), wo_img the out projection's weight in the w2 image format.  See include/vsde_hip.h."""
    lib = load(); dev = _require_hip(x, attn, glog, wo_img, ga, sc, sh, gm, w1_img, w2_img, b1_img)
    B, N, C = x.shape
    if not (x.is_contiguous() and attn.is_contiguous() and x.dtype == torch.bfloat16 and attn.dtype == torch.bfloat16 and attn.numel() == x.numel()):
        raise ValueError("mlp_attn_block_fwd: x and attn must be contiguous bf16 tensors of one size")
    if not (glog.ndim == 2 and glog.shape == (B * N, 64) and glog.dtype == torch.bfloat16 and glog.stride(1) == 1):
        raise ValueError("mlp_attn_block_fwd: glog must be [B*N, 64] bf16 with unit column stride")
    mp = _mod_pitch(C, ga, sc, sh, gm, sn, hs)
    tok = torch.empty_like(x)
    hnext = torch.empty_like(x) if sn is not None else None
    with torch.cuda.device(dev):
        _call(lib.vsde_mlp_attn_block_fwd_bf16, _ptr(x), _ptr(attn), _ptr(glog), _i64(glog.stride(0)), _ptr(wo_img), _ptr(bo), _ptr(ga), _ptr(sc),
              _ptr(sh), _ptr(gm), _ptr(sn), _ptr(hs), _i64(mp), ctypes.c_int(N), ctypes.c_double(eps), ctypes.c_double(eps_next), _ptr(w1_img),
              _ptr(w2_img), _ptr(b1_img), _ptr(b2), _ptr(tok), _ptr(hnext), _i64(B * N), ctypes.c_int(C), ctypes.c_int(H), _stream(dev))
    return tok, hnext


def linear_deep256(x: torch.Tensor, w_img: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """y [M,256] = x [M,K] W^T (+ bias) for K % 64 == 0, K >= 256: ``w_img`` = W as K/16 k-step images [K/16, 2, 256, 8] bf16
    (``fused.DeepImage``).  See include/vsde_hip.h."""
    lib = load(); dev = _require_hip(x, w_img)
    x, ldx = _rows2d(x)
    M, K = x.shape
    if w_img.dtype != torch.bfloat16 or w_img.numel() != 256 * K or not w_img.is_contiguous():
        raise ValueError("linear_deep256: w_img must be a contiguous bf16 image of 256 x K elements")
    y = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_deep256_bf16, _ptr(x), _i64(ldx), _ptr(w_img), _ptr(bias), _ptr(y), _i64(256), _i64(M), ctypes.c_int(K), _stream(dev))
    return y


def mlp_bwd_image_bytes(C: int) -> int:
    """Bytes per pair tile (32 hidden units) of the fused MLP backward's weight image; 0 = width not built."""
    return int(load().vsde_mlp_bwd_image_bytes(ctypes.c_int(C)))


def mlp_bwd(dy: torch.Tensor, u: torch.Tensor, img: torch.Tensor, H: int):
    """Fused SwiGLU MLP backward (csrc/vsde_mlp.hip): dy [M,C], saved u [M,2H] (interleaved layout) -> (du [M,2H], dx [M,C])."""
    lib = load(); dev = _require_hip(dy, u, img)
    dy, lddy = _rows2d(dy); u, ldu = _rows2d(u)
    M, C = dy.shape
    du = torch.empty(M, 2 * H, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        _call(lib.vsde_mlp_bwd_bf16, _ptr(dy), _i64(lddy), _ptr(u), _i64(ldu), _ptr(img), _ptr(du), _i64(2 * H), _ptr(dx), _i64(C), _i64(M),
              ctypes.c_int(C), ctypes.c_int(H), _stream(dev))
    return du, dx


def linear_gate_bwd(dy: torch.Tensor, w_t: torch.Tensor, og: torch.Tensor, s: torch.Tensor, dgate: torch.Tensor, tokens: int):
    """Input gradient of the attention output projection fused with the gate backward: dy [M,K] (gradient of the projection
    output), w_t [H*64, K] (its weight transposed), og [B,N,H,64] the merged gated rows, s / dgate row-pitched [M, >=64] (gate
    factors in, gate-logit gradient out).  Returns (dattn [B,N,H,64], delta [B,H,N] fp32) like ``gate_bwd_delta``."""
    lib = load(); dev = _require_hip(dy, w_t, og, s, dgate)
    dy, ldy = _rows2d(dy); s, lds = _rows2d(s); dgate, ldd = _rows2d(dgate)
    B, N, H, d = og.shape
    M, K = dy.shape
    if d != 64 or N != tokens or M != B * N or tuple(w_t.shape) != (H * 64, K) or not (og.is_contiguous() and w_t.is_contiguous()):
        raise ValueError("linear_gate_bwd: inconsistent shapes")
    dattn = torch.empty_like(og)
    delta = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_gate_bwd_bf16, _ptr(dy), _i64(ldy), _ptr(w_t), _ptr(og), _ptr(s), _i64(lds), _ptr(dattn), _ptr(dgate),
              _i64(ldd), _ptr(delta), _i64(M), ctypes.c_int(K), ctypes.c_int(H), ctypes.c_int(N), _stream(dev))
    return dattn, delta


def attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, vdiff, lam, dv0, dv_extra, dy, scale: float):
    """Attention backward whose epilogues apply the RoPE / RMS-norm / value-mix backward and write the q, k, v columns of ``dy``
    ([B*N, >= 192 H] bf16, the gradient of the [q | k | v | gate] projection).  ``dv0``: existing buffer to accumulate the
    residual-value gradient into (a new one is made when ``vdiff`` is given and ``dv0`` is None).  Returns (dv0, dlam)."""
    lib = load(); dev = _require_hip(dattn, q, k, v, lse, delta, rinv, dy)
    B, N, H, d = q.shape
    dy2, ldy = _rows2d(dy)
    accumulate = dv0 is not None
    parts = None
    if vdiff is not None:
        if dv0 is None:
            dv0 = torch.empty_like(v)
        parts = torch.empty(int(lib.vsde_attention_bwd_fused_partials(_i64(B), ctypes.c_int(N), ctypes.c_int(H))), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _call(lib.vsde_attention_bwd_fused_bf16, _ptr(dattn), _ptr(q), _ptr(k), _ptr(v), _ptr(lse), _ptr(delta), _ptr(rinv), _ptr(cos),
              _ptr(sin), _ptr(wq), _ptr(wk), _ptr(vdiff), _ptr(lam if vdiff is not None else None),
              _ptr(dv0 if vdiff is not None else None), ctypes.c_int(int(accumulate)), _ptr(dv_extra), _ptr(dy2), _i64(ldy), _ptr(parts),
              _i64(B), ctypes.c_int(N), ctypes.c_int(H), ctypes.c_double(scale), _stream(dev))
    return (dv0 if vdiff is not None else None), (parts.sum() if parts is not None else None)


def linear_gated_bf16(attn: torch.Tensor, gate: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]):
    """y [M,N] = (attn [M,K] * sigmoid(gate [M,64] broadcast over the heads)) w^T + bias: gate_merge folded into the GEMM."""
    lib = load(); dev = _require_hip(attn, gate, w)
    attn, ldx = _rows2d(attn); gate, ldg = _rows2d(gate)
    M, K = attn.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_gated_bf16, _ptr(attn), _i64(ldx), _ptr(gate), _i64(ldg), _ptr(w), _ptr(bias), _ptr(y), _i64(N), _i64(M),
              ctypes.c_int(N), ctypes.c_int(K), _stream(dev))
    return y


def linear_swiglu_bwd_bf16(dy: torch.Tensor, w_t: torch.Tensor, u: torch.Tensor):
    """du [M,2H] (interleaved, like u) from dy [M,K] and w_t [H,K] = the SwiGLU output projection transposed: the product
    ds = dy w_t^T never leaves the registers."""
    lib = load(); dev = _require_hip(dy, w_t, u)
    dy, ldx = _rows2d(dy)
    M, K = dy.shape
    H = w_t.shape[0]
    du = torch.empty(M, 2 * H, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        _call(lib.vsde_linear_bf16, _ptr(dy), _i64(ldx), _ptr(w_t), None, _ptr(du), _i64(2 * H), _i64(M), ctypes.c_int(H),
              ctypes.c_int(K), ctypes.c_int(EPI_SWIGLU_BWD), None, _i64(0), _ptr(u), _i64(u.stride(0)), _stream(dev))
    return du
