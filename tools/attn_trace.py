#!/usr/bin/env python3
"""Per-phase cycle sums of workgroup 0 of the persistent attention forward (csrc/vsde_attn.hip, vsde_attn_debug_trace), LV dims:
    python tools/attn_trace.py [N]"""
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
trace = torch.zeros(12, 5, device="cuda:0", dtype=torch.int64)
lib = _hip.load()
lib.vsde_attn_debug_trace(ctypes.c_void_p(trace.data_ptr()))
_hip.attention_fwd(q, k, v, 0.125)
torch.cuda.synchronize()
lib.vsde_attn_debug_trace(None)
names = ["stage K / V (request, barriers, LDS commit)", "block prologue (q fragments, norms)", "tile loop", "epilogue"]
print(f"N = {N}: cycles per (batch, head) pair, workgroup 0")
for w in range(12):
    n = max(int(trace[w, 4]), 1)
    print(f"wave {w:2d}: " + " | ".join(f"{nm} {int(trace[w, k]) / n:.0f}" for k, nm in enumerate(names)) + f" | total {int(trace[w, :4].sum()) / n:.0f} ({n} pairs)")
