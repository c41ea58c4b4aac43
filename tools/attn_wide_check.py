#!/usr/bin/env python3
"""Prototype check: the dq kernel with 64 queries per wave (VSDE_ATTN_DQ_WIDE=1, csrc/vsde_attn.hip::attn_bwd_dq_wide_kernel, unfused API)
against attn_bwd_dq_kernel<false>: bit identity of dq / dk / dv and the time of the whole backward (dq + dk/dv; the dk/dv launch is the
same kernel on both sides, so the difference is the dq kernel's).    python tools/attn_wide_check.py [B]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip

B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), 4
def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for N in (512, 448, 401, 384, 256):
    g = torch.Generator().manual_seed(N)
    R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
    q, k, v, go = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
    o, lse = _hip.attention_fwd(q, k, v, 0.125)
    res, tt = {}, {"0": [], "1": []}
    for rep in range(3):   # alternating: the first timed loop after an idle gap runs at ramping clocks
        for mode in ("0", "1"):
            os.environ["VSDE_ATTN_DQ_WIDE"] = mode
            res[mode] = _hip.attention_bwd(go, q, k, v, o, lse, 0.125)
            tt[mode].append(timeit(lambda: _hip.attention_bwd(go, q, k, v, o, lse, 0.125)))
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(res["0"], res["1"]))
    print(f"N = {N}: backward (dq + dk/dv) " + " | ".join(f"{a:6.1f} -> {b:6.1f}" for a, b in zip(tt["0"], tt["1"])) + f" us; bit-identical {same}")

# ---- the fused backward of the training step (projection-side epilogues), realistic operands (QK-normed, rotated q / k), alternating
print("fused backward at the LV shape (512 x 401 x 4 x 64), wide dq off / on alternately:")
B, N = 512, 401
M = B * N
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
x, w, bias = R(M, 256), R(832, 256) * 0.06, R(832) * 0.1
cos, sin = torch.cos(torch.rand(N, 32, generator=g) * 6).to("cuda:0"), torch.sin(torch.rand(N, 32, generator=g) * 6).to("cuda:0")
wq = torch.ones(64, device="cuda:0"); wk = torch.ones(64, device="cuda:0")
v0 = R(M, 256); lam = torch.tensor([0.4], device="cuda:0")
dout = R(B, N, H, 64)
dy = torch.empty(M, 832, device="cuda:0", dtype=torch.bfloat16)
acc = torch.zeros(B, N, H, 64, device="cuda:0", dtype=torch.bfloat16)
q, k, v, glog, rinv, vdiff = _hip.linear_qknorm_bf16(x, w, bias, H, N, cos, sin, wq, wk, v0, lam, 1e-6, save=True)
sh = (B, N, H, 64)
q, k, v = q.view(sh), k.view(sh), v.view(sh)
og, lse = _hip.attention_fwd_gated(q, k, v, glog, 0.125)
dattn, delta = _hip.gate_bwd_delta(dout, og, glog, dy[:, 768:])
run = lambda: _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, None, None, None, None, dy, 0.125)
outs = {}
for rep in range(4):
    for mode in ("0", "1"):
        os.environ["VSDE_ATTN_DQ_WIDE"] = mode
        tt = timeit(run, 20)
        outs[mode] = dy.clone()
        print(f"  wide = {mode}: {tt:7.1f} us")
torch.cuda.synchronize()
print("  dy bit-identical:", torch.equal(outs["0"], outs["1"]), " max |diff|", (outs["0"].float() - outs["1"].float()).abs().max().item())
