#!/usr/bin/env python3
"""One shape of the encoder GEMM kernels, timed with HIP events (ablation / profiling driver).
    python tools/linear_probe.py N K [iters]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = 205312
x = torch.randn(M, K, device="cuda:0").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda:0") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda:0").to(torch.bfloat16)
y = torch.empty(M, N, device="cuda:0", dtype=torch.bfloat16)
for _ in range(3):
    _hip.linear_bf16(x, w, b, out=y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(iters):
    _hip.linear_bf16(x, w, b, out=y)
e1.record(); torch.cuda.synchronize()
print(f"N={N} K={K} dbg={os.environ.get('VSDE_LIN_DEBUG', '0')}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us")
