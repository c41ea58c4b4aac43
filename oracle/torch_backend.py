"""TEST INFRASTRUCTURE ONLY -- CPU operator backend built on the C oracle.

Implements the same four operator entry points as ``viforsdes_amd._hip`` for CPU torch tensors so
that tests (and bench.py's ``cpu_baseline`` leg) can run the host logic -- autograd wiring, trainer,
data-parallel step over gloo -- without a GPU:

    from viforsdes_amd.kernels.backend import set_backend
    from oracle.torch_backend import OracleBackend
    set_backend(OracleBackend())        # tests / cpu baseline only

The shipped package never installs it."""
from __future__ import annotations

import numpy as np
import torch

from . import vsde_oracle as vo


def _np(t, dtype):
    return np.ascontiguousarray(t.detach().to(torch.float64 if dtype == np.float64 else torch.float32).cpu().numpy())


class OracleBackend:
    def __init__(self, dtype=np.float32) -> None:
        self.dtype = dtype
        self.tdtype = torch.float64 if dtype == np.float64 else torch.float32

    def _t(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.tdtype)

    def head_forward(self, x0, ctx, theta, eps, ws, time_step, save, diag_min=vo.DIAG_MIN):
        w = vo.HeadWeights(*[_np(t, self.dtype) for t in ws])
        f = vo.head_forward(_np(x0, self.dtype), _np(ctx, self.dtype), _np(theta, self.dtype), _np(eps, self.dtype),
                            w, float(time_step), bool(save), self.dtype, diag_min)
        return (self._t(f.paths), self._t(f.means), self._t(f.chol),
                self._t(f.chol_raw) if save else None, self._t(f.acts) if save else None)

    def head_backward(self, g_paths, g_means, g_chol, ctx, theta, eps, paths, chol_raw, acts, ws, time_step,
                      diag_min=vo.DIAG_MIN, context_grad_out=None):
        w = vo.HeadWeights(*[_np(t, self.dtype) for t in ws])
        f = vo.FwdResult(_np(paths, self.dtype), None, None, _np(chol_raw, self.dtype), _np(acts, self.dtype))
        g = vo.head_backward(_np(g_paths, self.dtype), _np(g_means, self.dtype), _np(g_chol, self.dtype),
                             _np(ctx, self.dtype), _np(theta, self.dtype), _np(eps, self.dtype), f, w, float(time_step),
                             self.dtype, diag_min)
        out = [self._t(a) for a in g]
        if context_grad_out is not None:  # same contract as _hip.head_backward: first T steps, caller's dtype
            T = out[1].shape[1]
            context_grad_out[:, :T].copy_(out[1])
            out[1] = context_grad_out
        return tuple(out)

    def elbo_path_terms(self, z, x, means, chol, drift, diffusion, positive_dims, time_step):
        out = vo.elbo_path_terms(*[_np(t, self.dtype) for t in (z, x, means, chol, drift, diffusion)],
                                 list(positive_dims), float(time_step), self.dtype)
        return tuple(self._t(a) for a in out)

    def elbo_path_terms_bwd(self, z, x, means, chol, drift, diffusion, positive_dims, time_step, g_sde, g_gen, g_jac):
        out = vo.elbo_path_terms_bwd(*[_np(t, self.dtype) for t in (z, x, means, chol, drift, diffusion)],
                                     list(positive_dims), float(time_step), _np(g_sde, self.dtype),
                                     _np(g_gen, self.dtype), _np(g_jac, self.dtype), self.dtype)
        return tuple(self._t(a) for a in out)
