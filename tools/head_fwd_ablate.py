#!/usr/bin/env python3
"""Timing-only ablations of the multi-path MFMA forward (VSDE_MP_FWD_ABL bits: 1 = layer 0 does not store its saved activations, 2 = layer 1
does not, 4 = no output stores) at the LV head dims; wrong results.   python tools/head_fwd_ablate.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip
from head_mp_check import inputs
T, S, C, P, H, L = 400, 2, 256, 3, 64, 2
for B in (512,):
    ws, x0, ctx, theta, eps = inputs(B, T, S, C, P, H, L, 3)
    d = lambda t: t.to("cuda:0")
    wd = [d(w) for w in ws]; x0, ctx, theta, eps = d(x0), d(ctx), d(theta), d(eps)
    for save in (True, False):
        _hip.profile_enable(True); ms = []
        for i in range(8):
            _hip.head_forward(x0, ctx[:, :-1], theta, eps, wd, 0.1, save)
            if i >= 2: ms.append(_hip.profile_elapsed_ms(0))
        _hip.profile_enable(False)
        print("ABL", os.environ.get("VSDE_MP_FWD_ABL", "0"), "B", B, "train" if save else "eval", f"{1e3 * sum(ms) / len(ms):.0f} us")
