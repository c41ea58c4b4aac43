#!/usr/bin/env python3
"""The rows-kernel GEMMs of a SiT block at the LV shape (M = 205,312, K = 256), HIP-event timings: plain N = 832 | [q k v gate] projection
with the QK-norm / RoPE epilogue (no-grad and training forms) | SwiGLU input projection (u + s) | SwiGLU backward (ds -> du).
    python tools/rows_probe.py            (VSDE_LIN_DEBUG=1: the same kernels without their output stores -- timing only)"""
import os as _os
if _os.environ.get("VSDE_LIN_DEBUG"):   # the ablation switch exists only in the tools' library; otherwise the shipped library is timed
    _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip

dev = torch.device("cuda:0")
B, N_tok, C, heads, H = 512, 401, 256, 4, 704
M = B * N_tok
g = torch.Generator().manual_seed(0)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
x = R(M, C).to(torch.bfloat16)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dbg = os.environ.get("VSDE_LIN_DEBUG", "0")
w832, b832 = R(832, C, sc=C ** -0.5).to(torch.bfloat16), R(832).to(torch.bfloat16)
y832 = torch.empty(M, 832, device=dev, dtype=torch.bfloat16)
print(f"dbg={dbg}  plain [M,256] x [832,256]^T                      {timeit(lambda: _hip.linear_bf16(x, w832, b832, out=y832)):7.1f} us   (HBM floor ~{(M * C * 2 + M * 832 * 2) / 5.25e6:.0f} us)")
cos, sin = torch.rand(N_tok, 32, device=dev), torch.rand(N_tok, 32, device=dev)
wq, wk = torch.ones(64, device=dev), torch.ones(64, device=dev)
v0 = R(M, heads * 64).to(torch.bfloat16)
lam = torch.tensor([0.37], device=dev)
print(f"dbg={dbg}  qk-norm projection, no-grad form                  {timeit(lambda: _hip.linear_qknorm_bf16(x, w832, b832, heads, N_tok, cos, sin, wq, wk, v0, lam, 1e-6)):7.1f} us")
print(f"dbg={dbg}  qk-norm projection, training form (rinv, vdiff)   {timeit(lambda: _hip.linear_qknorm_bf16(x, w832, b832, heads, N_tok, cos, sin, wq, wk, v0, lam, 1e-6, save=True)):7.1f} us")
print(f"dbg={dbg}  qk-norm projection, no value mix                  {timeit(lambda: _hip.linear_qknorm_bf16(x, w832, b832, heads, N_tok, cos, sin, wq, wk, None, None, 1e-6)):7.1f} us")
w1, b1 = R(2 * H, C, sc=C ** -0.5).to(torch.bfloat16), R(2 * H).to(torch.bfloat16)
print(f"dbg={dbg}  SwiGLU input projection (u + s)                   {timeit(lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=True)):7.1f} us   (HBM floor ~{(M * C * 2 + M * 3 * H * 2) / 5.25e6:.0f} us)")
print(f"dbg={dbg}  SwiGLU input projection (s only)                  {timeit(lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=False)):7.1f} us")
u, _ = _hip.linear_swiglu_bf16(x, w1, b1, want_u=True)
w2t = R(H, C, sc=H ** -0.5).to(torch.bfloat16)
dy = R(M, C).to(torch.bfloat16)
print(f"dbg={dbg}  SwiGLU backward (ds -> du)                        {timeit(lambda: _hip.linear_swiglu_bwd_bf16(dy, w2t, u)):7.1f} us   (HBM floor ~{(M * C * 2 + M * 4 * H * 2) / 5.25e6:.0f} us)")
