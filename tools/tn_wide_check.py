#!/usr/bin/env python3
"""Weight gradients of the fused head at the LV size under the switches of vsde_tn_wide.hip (ablation library): each combination runs in
its own process (a faulting kernel must not take the others down), writes its gradients, and is compared with the fp32-MFMA, tile-major
form.   python tools/tn_wide_check.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:   # worker: run once, save the gradients
    os.environ.setdefault("VSDE_HIP_LIB", os.path.join(ROOT, "viforsdes_amd", "libvsde_hip_abl.so"))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from viforsdes_amd import _hip
    from head_mp_check import inputs
    B, T, S, C, P, H, L = 512, 400, 2, 256, 3, 64, 2
    ws, x0, ctx, theta, eps = inputs(B, T, S, C, P, H, L, 3)
    d = lambda t: t.to("cuda:0")
    wd = [d(w) for w in ws]; x0, ctx, theta, eps = d(x0), d(ctx), d(theta), d(eps)
    g = torch.Generator(device="cpu").manual_seed(5)
    gp, gm, gl = (d(torch.randn(*s, generator=g)) for s in ((B, T + 1, S), (B, T, S), (B, T, S, S)))
    fo = _hip.head_forward(x0, ctx[:, :-1], theta, eps, wd, 0.1, True)
    out = _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, fo[0], fo[3], fo[4], wd, 0.1)
    torch.cuda.synchronize()
    torch.save([o.float().cpu() for o in out], sys.argv[1])
    sys.exit(0)
import torch
combos = ["VSDE_TW_SPLIT=0 VSDE_TW_INTERLEAVE=0", "VSDE_TW_SPLIT=0 VSDE_TW_INTERLEAVE=1", "VSDE_TW_SPLIT=1 VSDE_TW_INTERLEAVE=0",
          "VSDE_TW_SPLIT=1 VSDE_TW_INTERLEAVE=1"]
ref = None
with tempfile.TemporaryDirectory() as td:
    for i, c in enumerate(combos):
        env = dict(os.environ); env.update(kv.split("=") for kv in c.split())
        f = os.path.join(td, f"g{i}.pt")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), f], env=env, timeout=120, capture_output=True, text=True)
            rc = r.returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
        if rc != 0:
            print(f"{c}: FAILED rc={rc}"); print((r.stderr or "")[-600:] if rc != "timeout" else ""); continue
        gs = torch.load(f)
        if ref is None: ref = gs
        err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(gs, ref))
        print(f"{c}: ok, max relative-to-peak difference from the first form {err:.3e}")
