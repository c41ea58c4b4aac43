#!/usr/bin/env python3
"""Per-phase cycle sums of workgroup 0 of the fused MLP backward (csrc/vsde_mlp.hip::mlp_bwd_kernel, vsde_mlp_debug_trace):
    python tools/mlp_bwd_trace.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402
from viforsdes_amd.primitives import fused  # noqa: E402

dev, M, C, hreal, H = "cuda:0", 205312, 256, 682, 704
P = lambda *s, sc=1.0: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)
w_in, b_in, w_out, b_out = P(2 * hreal, C, sc=C ** -0.5), P(2 * hreal), P(C, hreal, sc=hreal ** -0.5), P(C)
x = torch.randn(M, C, device=dev).to(torch.bfloat16)
dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
pin, pout = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=True)
img = fused.MlpBwdImages(pin, pout, H).operand()
u, _ = _hip.linear_swiglu_bf16(x, *pin.operands(), want_u=True)
trace = torch.zeros(4, 8, device=dev, dtype=torch.int64)
for _ in range(2):
    _hip.mlp_bwd(dy, u, img, H)
_hip.load().vsde_mlp_debug_trace(ctypes.c_void_p(trace.data_ptr()))
_hip.mlp_bwd(dy, u, img, H)
torch.cuda.synchronize()
_hip.load().vsde_mlp_debug_trace(None)
names = ["u -> staging + next u issue", "pieces + E' (du math)", "du write-back + stores", "G3 (dx)", "(end)", "barrier", "tile regs -> LDS + next tile issue", "G1' (ds)"]
tp = H // 32
for w in range(4):
    print(f"wave {w}: " + " | ".join(f"{n} {int(trace[w, k]) / tp:.0f}" for k, n in enumerate(names)) + f" | total {int(trace[w, :8].sum()) / tp:.0f} cycles per pair tile")
