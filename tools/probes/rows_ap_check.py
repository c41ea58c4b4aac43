#!/usr/bin/env python3
"""Every epilogue of the rows GEMM at the LV row count (and a ragged one): time + a digest of the outputs.  Run once per setting of
VSDE_ROWS_AP / VSDE_ROWS_NW (read once per process) and compare the digests: the eight-wave anti-phase workgroups must be
bit-identical to the four-wave ones.
    VSDE_ROWS_AP=0 python tools/probes/rows_ap_check.py ; VSDE_ROWS_AP=31 python tools/probes/rows_ap_check.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from viforsdes_amd import _hip  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(5)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bf = lambda t: t.to(torch.bfloat16)


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        if t is not None:
            h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("VSDE_ROWS_AP =", os.environ.get("VSDE_ROWS_AP"), " VSDE_ROWS_NW =", os.environ.get("VSDE_ROWS_NW"))
for B, N in ((512, 401), (173, 401)):
    M, K, H, heads = B * N, 256, 704, 4
    x = bf(rn(M, K))
    w = bf(rn(256, K, sc=K ** -0.5)); b = bf(rn(256))
    w1 = bf(rn(2 * H, K, sc=K ** -0.5)); b1 = bf(rn(2 * H))
    w2t = bf(rn(H, K, sc=H ** -0.5))
    dy = bf(rn(M, K))
    wqkv = bf(rn(3 * 256 + 64, K, sc=K ** -0.5)); bqkv = bf(rn(3 * 256 + 64))
    cos, sin = rn(N, 32).cos(), rn(N, 32).sin()
    wq, wk = rn(64).abs() + 0.5, rn(64).abs() + 0.5
    v0 = bf(rn(M, 256)); lam = torch.tensor([0.7], device=dev)
    og = bf(rn(B, N, heads, 64)); sg = bf(torch.sigmoid(rn(M, 64))); dgate = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
    wo_t = bf(rn(256, K, sc=K ** -0.5))
    u, _ = _hip.linear_swiglu_bf16(x, w1, b1)
    ops = {
        "plain": lambda: (_hip.linear_bf16(x, w, b),),
        "swiglu train": lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=True),
        "swiglu nograd": lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=False),
        "swiglu bwd": lambda: (_hip.linear_swiglu_bwd_bf16(dy, w2t, u),),
        "qknorm train": lambda: _hip.linear_qknorm_bf16(x, wqkv, bqkv, heads, N, cos, sin, wq, wk, v0, lam, 1e-6, save=True),
        "qknorm nograd": lambda: _hip.linear_qknorm_bf16(x, wqkv, bqkv, heads, N, cos, sin, wq, wk, v0, lam, 1e-6, save=False),
        "gate bwd": lambda: _hip.linear_gate_bwd(dy, wo_t, og, sg, dgate, N) + (dgate,),
    }
    for name, fn in ops.items():
        out = fn()
        print(f"M={M:7d} {name:14s} {timeit(fn):7.1f} us  {digest(*out)}")
