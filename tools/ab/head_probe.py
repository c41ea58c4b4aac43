#!/usr/bin/env python3
"""The fused head alone at the LV benchmark size (B=512, T=400, S=2, C=256 bf16 context, H=64, L=2): forward (training
variant) + backward, a few repetitions -- the driver for rocprofv3 kernel-trace / PMC passes over the head kernels.
    python tools/head_probe.py [reps] [B] [synthetic]      (synthetic: B=256, T=1000, S=8, C=512, P=16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T, S, C, P, H, L = 400, 2, 256, 3, 64, 2
if len(sys.argv) > 3 and sys.argv[3] == "synthetic":
    T, S, C, P = 1000, 8, 512, 16
NO = S + S * (S + 1) // 2
obias = torch.zeros(NO)
for k in range(S):
    obias[S + k * (k + 3) // 2] = 1.0
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(3)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
ws = [rn(3 * H, S + C + P, sc=0.08), rn(3 * H, H, sc=0.12), rn(3 * H, sc=0.1), rn(3 * H, sc=0.1),
      rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, H, sc=0.12), rn(L - 1, 3 * H, sc=0.1), rn(L - 1, 3 * H, sc=0.1),
      rn(NO, H, sc=0.1), obias.to(dev)]
x0, ctx, theta, eps = rn(B, S), rn(B, T + 1, C).to(torch.bfloat16), rn(B, P).abs(), rn(B, T, S)
gp, gm, gl = rn(B, T + 1, S), rn(B, T, S), rn(B, T, S, S)
gctx = torch.empty(B, T + 1, C, device=dev, dtype=torch.bfloat16)
dt = 0.1 if S == 2 else 0.01
_hip.profile_enable(True)
ms = {k: [] for k in range(7)}
for i in range(reps + 2):
    out = _hip.head_forward(x0, ctx[:, :-1], theta, eps, ws, dt, True)
    _hip.head_backward(gp, gm, gl, ctx[:, :-1], theta, eps, out[0], out[3], out[4], ws, dt, context_grad_out=gctx)
    if i >= 2:
        for k in ms:
            ms[k].append(_hip.profile_elapsed_ms(k))
torch.cuda.synchronize()
names = ["serial fwd", "serial bwd", "whole fwd", "whole bwd", "proj gemm", "grad_ctx gemm", "weight-grad reduction"]
print(" | ".join(f"{n} {1e3 * sum(v) / len(v):.0f} us" for n, v in zip(names, ms.values())))
