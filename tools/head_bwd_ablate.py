#!/usr/bin/env python3
"""Timing-only ablations of the multi-path MFMA reverse-time sweep (VSDE_MP_BWD_ABL bits: 1 no stores, 2 no loads in the loop, 4 no split
arithmetic, 8 no matrix products) at the LV head dims, B = 512, 4 paths per workgroup.  Results of the ablated variants are wrong."""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip
from head_mp_check import inputs
B, T, S, C, P, H, L = 512, 400, 2, 256, 3, 64, 2
ws, x0, ctx, theta, eps = inputs(B, T, S, C, P, H, L, 3)
d = lambda t: t.to("cuda:0")
wd = [d(w) for w in ws]; x0, ctx, theta, eps = d(x0), d(ctx), d(theta), d(eps)
_hip.debug_head_mp(4)
fo = _hip.head_forward(x0, ctx[:, :-1], theta, eps, wd, 0.1, True)
gp, gm, gl = torch.randn(B, T + 1, S, device="cuda:0"), torch.randn(B, T, S, device="cuda:0"), torch.randn(B, T, S, S, device="cuda:0")
_hip.profile_enable(True)
ms = []
for i in range(6):
    _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, fo[0], fo[3], fo[4], wd, 0.1)
    if i >= 2: ms.append(_hip.profile_elapsed_ms(1))
print("ABL", os.environ.get("VSDE_MP_BWD_ABL", "0"), f"backward sweep {1e3 * sum(ms) / len(ms):7.0f} us")
