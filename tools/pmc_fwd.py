"""Runs only the LV-size head forward (training variant) + backward a few times; used under
rocprofv3 --pmc to collect FETCH_SIZE / WRITE_SIZE of the serial kernels (see profiles/)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip
dev = torch.device("cuda:0")
B, T, S, C, P, H, L = 512, 400, 2, 256, 3, 64, 2
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
NO = S + S * (S + 1) // 2
ws = [rn(3*H, S+C+P, sc=.08), rn(3*H, H, sc=.12), rn(3*H, sc=.1), rn(3*H, sc=.1), rn(L-1, 3*H, H, sc=.12), rn(L-1, 3*H, H, sc=.12),
      rn(L-1, 3*H, sc=.1), rn(L-1, 3*H, sc=.1), rn(NO, H, sc=.1), torch.ones(NO).to(dev)]
x0, ctx, theta, eps = rn(B, S), rn(B, T+1, C).to(torch.bfloat16)[:, :-1], rn(B, P).abs(), rn(B, T, S)
gp, gm, gl = rn(B, T+1, S), rn(B, T, S), rn(B, T, S, S)
for _ in range(4):
    out = _hip.head_forward(x0, ctx, theta, eps, ws, 0.1, True)
    _hip.head_backward(gp, gm, gl, ctx, theta, eps, out[0], out[3], out[4], ws, 0.1)
torch.cuda.synchronize()
