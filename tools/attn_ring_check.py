#!/usr/bin/env python3
"""The ring forward (VSDE_ATTN_RING=1) against the LDS-resident forward: same arithmetic in the same order, so o and lse must be
bit-identical.  Run once without the variable (writes gpurun_out/attn_ring_ref.npz), once with it (compares and times).
    python tools/attn_ring_check.py; VSDE_ATTN_RING=1 python tools/attn_ring_check.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip

ring = os.environ.get("VSDE_ATTN_RING", "0") != "0"
path = "gpurun_out/attn_ring_ref.npz"
ref = dict(np.load(path)) if ring and os.path.exists(path) else None   # (no reference: timing only)
dev, H, out, bad = "cuda:0", 4, {}, 0
for B, N in ((512, 401), (512, 385), (512, 416), (300, 300), (256, 257)):
    g = torch.Generator().manual_seed(N)
    R = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16)
    q, k, v = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
    gate = torch.sigmoid(R(B * N, 64).float()).to(torch.bfloat16)
    for name, fn in (("gated", lambda: _hip.attention_fwd_gated(q, k, v, gate, 0.125)), ("plain", lambda: _hip.attention_fwd(q, k, v, 0.125))):
        o, lse = fn()
        torch.cuda.synchronize()
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        key = f"{B}x{N}_{name}"
        out[key + "_o"], out[key + "_lse"] = o.float().cpu().numpy(), lse.cpu().numpy()
        msg = f"{key:16s} {us:7.1f} us"
        if ref is not None:
            do, dl = int((out[key + "_o"] != ref[key + "_o"]).sum()), int((out[key + "_lse"] != ref[key + "_lse"]).sum())
            nan = int(np.isnan(out[key + "_o"]).sum())
            bad += (do > 0) + (dl > 0)
            msg += f"   o differs in {do} of {o.numel()} (nan {nan}), lse in {dl}: {'bit-identical' if do == 0 and dl == 0 else 'MISMATCH'}"
        print(msg, flush=True)
if ring and ref is not None:
    print("RING CHECK", "PASS" if bad == 0 else f"FAIL ({bad})")
elif not ring:
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(path, **out)
    print("reference written")
