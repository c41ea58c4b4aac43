#!/usr/bin/env python3
"""Per-phase cycle stamps of workgroup 0 of the fused MLP forward (VSDE_MLP_DEBUG=16 build of csrc/vsde_mlp.hip):
    VSDE_MLP_DEBUG=16 python tools/mlp_trace.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VSDE_MLP_DEBUG", "16")
from viforsdes_amd import _hip  # noqa: E402
from viforsdes_amd.primitives import fused  # noqa: E402

dev, M, C, hreal, H = "cuda:0", 205312, 256, 682, 704
T = H // 16
P = lambda *s, sc=1.0: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)
w_in, b_in, w_out, b_out = P(2 * hreal, C, sc=C ** -0.5), P(2 * hreal), P(C, hreal, sc=hreal ** -0.5), P(C)
x = torch.randn(M, C, device=dev).to(torch.bfloat16)
pin, pout = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=False)
w1, w2, b1 = fused.MlpImages(pin, pout, H).operands()
trace = torch.zeros(8, 2 * T + 2, 2, device=dev, dtype=torch.int64)
_hip.load().vsde_mlp_debug_trace(ctypes.c_void_p(trace.data_ptr()))
for _ in range(3):
    _hip.mlp_fwd(x, w1, w2, b1, pout.bias, H)
torch.cuda.synchronize()
tr = trace.cpu()
t0 = int(tr[tr > 0].min())
print("phase n: per wave (arrive at barrier, leave barrier) in cycles since the first stamp; waves 0-3 = group 0, 4-7 = group 1")
for n in range(0, 2 * T + 1):
    row = []
    for w in (0, 1, 4, 5):
        a, b = int(tr[w, n, 0]), int(tr[w, n, 1])
        row.append(f"w{w}: {a - t0 if a else -1:7d} {b - t0 if b else -1:7d}")
    if n < 24 or n > 2 * T - 6:
        print(f"n={n:3d}  " + "   ".join(row))
# mean phase durations in the steady state
import statistics
for w in (0, 4):
    busy = [int(tr[w, n, 0] - tr[w, n - 1, 1]) for n in range(8, 2 * T - 8)]
    wait = [int(tr[w, n, 1] - tr[w, n, 0]) for n in range(8, 2 * T - 8)]
    even = [b for n, b in zip(range(8, 2 * T - 8), busy) if n % 2 == 0]
    odd = [b for n, b in zip(range(8, 2 * T - 8), busy) if n % 2 == 1]
    print(f"wave {w}: work before even-n barriers {statistics.mean(even):.0f} cycles, before odd-n barriers {statistics.mean(odd):.0f}; "
          f"barrier wait {statistics.mean(wait):.0f}")
