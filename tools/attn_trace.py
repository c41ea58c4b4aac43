#!/usr/bin/env python3
"""Per-phase cycle sums of workgroup 0 of the persistent attention forward (csrc/vsde_attn.hip, vsde_attn_debug_trace), LV dims:
    python tools/attn_trace.py [N]          (VSDE_ATTN_FWD8=1: the eight-wave kernel)"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip
B, H = 512, 4
N = int(sys.argv[1]) if len(sys.argv) > 1 else 401
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
q, k, v = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
for _ in range(3):
    _hip.attention_fwd(q, k, v, 0.125)
f8 = os.environ.get("VSDE_ATTN_FWD8") == "1" and 384 < N <= 416   # the eight-wave kernel stamps other phases
trace = torch.zeros(8, 8, device="cuda:0", dtype=torch.int64) if f8 else torch.zeros(12, 5, device="cuda:0", dtype=torch.int64)
lib = _hip.load()
lib.vsde_attn_debug_trace(ctypes.c_void_p(trace.data_ptr()))
_hip.attention_fwd(q, k, v, 0.125)
torch.cuda.synchronize()
lib.vsde_attn_debug_trace(None)
names = ["stage K / V (request, barriers, LDS commit)", "block prologue (q fragments, norms)", "tile loop", "epilogue"]
nw = 12
if f8:
    names, nw = ["tile loops", "prologues", "waits for rows", "epilogues / partials", "first barrier", "commit + second barrier", "request issue"], 8
print(f"N = {N}: cycles per (batch, head) pair, workgroup 0")
for w in range(nw):
    n = max(int(trace[w, -1]), 1)
    print(f"wave {w:2d}: " + " | ".join(f"{nm} {int(trace[w, k]) / n:.0f}" for k, nm in enumerate(names)) + f" | total {int(trace[w, :-1].sum()) / n:.0f} ({n} pairs)")
