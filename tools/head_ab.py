#!/usr/bin/env python3
"""GRU time-stepping kernels at the LV head dims, 512 paths, dispatcher defaults: training forward, sampling forward, reverse sweep (us).
For same-box A/B of two library builds: tools/ab_lib.sh <other .so> python tools/head_ab.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip
from head_mp_check import inputs
B, T, S, C, P, H, L = 512, 400, 2, 256, 3, 64, 2
ws, x0, ctx, theta, eps = inputs(B, T, S, C, P, H, L, 3)
d = lambda t: t.to("cuda:0")
wd = [d(w) for w in ws]; x0, ctx, theta, eps = d(x0), d(ctx), d(theta), d(eps)
gp, gm, gl = torch.randn(B, T + 1, S, device="cuda:0"), torch.randn(B, T, S, device="cuda:0"), torch.randn(B, T, S, S, device="cuda:0")
out = []
for save in (True, False):
    _hip.profile_enable(True); ms = []
    for i in range(10):
        fo = _hip.head_forward(x0, ctx[:, :-1], theta, eps, wd, 0.1, save)
        if i >= 2: ms.append(_hip.profile_elapsed_ms(0))
    _hip.profile_enable(False)
    out.append(1e3 * sum(ms) / len(ms))
fo = _hip.head_forward(x0, ctx[:, :-1], theta, eps, wd, 0.1, True)
_hip.profile_enable(True); ms = []
for i in range(10):
    _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, fo[0], fo[3], fo[4], wd, 0.1)
    if i >= 2: ms.append(_hip.profile_elapsed_ms(1))
_hip.profile_enable(False)
print(f"train fwd {out[0]:6.1f} us | eval fwd {out[1]:6.1f} us | bwd {1e3 * sum(ms) / len(ms):6.1f} us")
