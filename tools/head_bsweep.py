#!/usr/bin/env python3
"""Batch sweep of the serial GRU kernels at the LV head dims (T=400, S=2, C=256 bf16 context, H=64, L=2): training forward,
no-grad (sampling) forward and backward, HIP events on the launch stream, priced with SURVEY 8(d)'s algorithmic bytes.
Shows where this design (4 waves per path, 2 paths per CU resident) saturates.   python tools/head_bsweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402

T, S, C, P, H, L = 400, 2, 256, 3, 64, 2
dev = "cuda:0"
ntril = S * (S + 1) // 2
fwd_b = 4 * (C // 2 + S + (2 * S + S * S + ntril) + 5 * L * H)
eval_b = 4 * (C // 2 + S + 2 * S + S * S)
bwd_b = 4 * (C // 2 + C // 2 + 4 * S + S * S + ntril + 5 * L * H)
print(f"bytes per path-step: train fwd {fwd_b}, eval fwd {eval_b}, bwd {bwd_b}; HBM peak 8000 GB/s")
print(f"{'B':>6} | {'train fwd us':>12} {'GB/s':>7} {'frac':>6} | {'eval fwd us':>11} {'GB/s':>7} {'frac':>6} | {'bwd us':>8} {'GB/s':>7} {'frac':>6} | paths/s (eval kernel)")
for B in (128, 256, 512, 1024, 2048, 4096, 8192):
    g = torch.Generator(device="cpu").manual_seed(3)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
          rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
          rn(S + 3, H, sc=0.1), torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0]).to(dev)]
    x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C).to(torch.bfloat16), rn(B, P).abs(), rn(B, T, S)
    gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
    gctx = torch.empty(B, T + 1, C, device=dev, dtype=torch.bfloat16)
    _hip.profile_enable(True)
    tf, te, tb = [], [], []
    for i in range(6):
        out = _hip.head_forward(x0, ctx[:, :-1], theta, eps, ws, 0.1, True)
        f = _hip.profile_elapsed_ms(0)
        _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, out[0], out[3], out[4], ws, 0.1, context_grad_out=gctx)
        b = _hip.profile_elapsed_ms(1)
        _hip.head_forward(x0, ctx[:, :-1], theta, eps, ws, 0.1, False)
        e = _hip.profile_elapsed_ms(0)
        if i >= 2:
            tf.append(f); te.append(e); tb.append(b)
    _hip.profile_enable(False)
    m = lambda v: sum(v) / len(v)
    gb = lambda by, ms: by * B * T / (ms * 1e-3) / 1e9
    f, e, b = m(tf), m(te), m(tb)
    print(f"{B:>6} | {1e3 * f:>12.0f} {gb(fwd_b, f):>7.0f} {gb(fwd_b, f) / 8000:>6.3f} | {1e3 * e:>11.0f} {gb(eval_b, e):>7.0f} {gb(eval_b, e) / 8000:>6.3f} | "
          f"{1e3 * b:>8.0f} {gb(bwd_b, b):>7.0f} {gb(bwd_b, b) / 8000:>6.3f} | {B / (e * 1e-3):>10.0f}")
    del out, ctx, gctx
    torch.cuda.empty_cache()
