#!/usr/bin/env python3
"""The fused attention backward with the ragged 13th block shared by four waves against the same kernels with the lone second round:
run once with VSDE_ATTN_SPLIT=0 (writes gpurun_out/attn_split_ref.npz) and once without (compares): the two differ only in the order
of fp32 partial sums, so every output must agree to ~1e-6 of its scale before bf16 rounding, i.e. in all but a few bf16 ulps.
    VSDE_ATTN_SPLIT=0 python tools/attn_split_check.py; python tools/attn_split_check.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip

ref_mode = os.environ.get("VSDE_ATTN_SPLIT", "1") == "0"
path = os.environ.get("VSDE_SPLIT_REF", "gpurun_out/attn_split_ref.npz")   # (tens of MB: the test suite points it at a temp dir)
ref = None if ref_mode else dict(np.load(path))
out, H, dev, bad = {}, 4, "cuda:0", 0
for N in (385, 386, 400, 401, 402, 403, 416):
    B = 6
    M = B * N
    g = torch.Generator().manual_seed(N)
    R = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16)
    x, w, b = R(M, 256), R(832, 256) * 0.06, R(832) * 0.1
    cos, sin = torch.cos(torch.rand(N, 32, generator=g) * 6).to(dev), torch.sin(torch.rand(N, 32, generator=g) * 6).to(dev)
    wq, wk = (1 + 0.1 * torch.randn(64, generator=g)).to(dev), (1 + 0.1 * torch.randn(64, generator=g)).to(dev)
    v0, lam, dout = R(M, 256), torch.tensor([0.4], device=dev), R(B, N, H, 64)
    q, k, v, glog, rinv, vdiff = _hip.linear_qknorm_bf16(x, w, b, H, N, cos, sin, wq, wk, v0, lam, 1e-6, save=True)
    sh = (B, N, H, 64)
    q, k, v = q.view(sh), k.view(sh), v.view(sh)
    og, lse = _hip.attention_fwd_gated(q, k, v, glog, 0.125)
    for mode in ("mix", "extra", "plain"):
        dy = torch.zeros(M, 832, device=dev, dtype=torch.bfloat16)
        acc = R(B, N, H, 64)
        dattn, delta = _hip.gate_bwd_delta(dout, og, glog, dy[:, 768:])
        if mode == "mix":
            res = _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, vdiff, lam, acc, None, dy, 0.125)
        elif mode == "extra":
            res = _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, None, None, None, acc, dy, 0.125)
        else:
            res = _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, None, None, None, None, dy, 0.125)
        torch.cuda.synchronize()
        items = {"dy": dy.float(), "acc": acc.float()}
        if isinstance(res, tuple) and res[1] is not None:
            items["dlam"] = res[1].float().reshape(-1)
        for name, t in items.items():
            key = f"{N}_{mode}_{name}"
            a = t.cpu().numpy()
            out[key] = a
            if ref is not None:
                r = ref[key]
                scale = np.abs(r).max() + 1e-30
                d = np.abs(a - r).max() / scale
                n_diff = int((a != r).sum())
                # a different fp32 summation order moves a value across a bf16 rounding boundary now and then: a few elements by one ulp
                ok = d < (2e-5 if name == "dlam" else 1e-2) and n_diff <= max(8, a.size // 200)
                bad += not ok
                print(f"{key:18s} max|diff|/max|ref| {d:.2e}  elements that differ {n_diff} of {a.size}  {'ok' if ok else 'MISMATCH'}")
if ref_mode:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    np.savez(path, **out)
    print("reference written:", path, len(out), "arrays")
else:
    print("SPLIT CHECK", "PASS" if bad == 0 else f"FAIL ({bad})")
    sys.exit(1 if bad else 0)
