#!/usr/bin/env python3
"""The attention forward alone at the LV dims (PMC / rocprof driver):  [VSDE_ATTN_RING=1] python tools/attn_fwd_probe.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip
B, N, H = 512, 401, 4
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
q, k, v = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    _hip.attention_fwd(q, k, v, 0.125)
torch.cuda.synchronize()
