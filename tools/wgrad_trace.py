#!/usr/bin/env python3
"""Per-phase cycle sums of workgroup 0 of the weight-gradient kernel (csrc/vsde_wgrad.hip, eight-wave TN = 256 form; vsde_wgrad_debug_trace):
    python tools/wgrad_trace.py [N K]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402

dev, M = "cuda:0", 205312
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1408, 256)
dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
trace = torch.zeros(8, 5, device=dev, dtype=torch.int64)
for _ in range(2):
    _hip.linear_wgrad(dy, x, True)
lib = _hip.load()
lib.vsde_wgrad_debug_trace(ctypes.c_void_p(trace.data_ptr()))
_hip.linear_wgrad(dy, x, True)
torch.cuda.synchronize()
lib.vsde_wgrad_debug_trace(None)
names = ["fragment reads + MFMAs", "staging registers -> LDS (+ bias sums)", "barrier", "loads for step s + 3 issued"]
print(f"dW[{N},{K}], M = {M}: cycles per 32-row step, workgroup 0")
for w in range(8):
    n = max(int(trace[w, 4]), 1)
    print(f"wave {w}: " + " | ".join(f"{nm} {int(trace[w, k]) / n:.0f}" for k, nm in enumerate(names)) + f" | total {int(trace[w, :4].sum()) / n:.0f} ({n} steps)")
