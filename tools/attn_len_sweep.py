#!/usr/bin/env python3
"""Attention core at the LV width (512 x 4 pairs, head_dim 64) over sequence lengths around the benchmark's 401 tokens: what the ragged
13th 32-token block costs each kernel (384 = twelve whole blocks, 416 = thirteen).   python tools/attn_len_sweep.py"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip

B, H = 512, 4
def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for N in (352, 384, 385, 401, 416):
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
    q, k, v = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
    t = timeit(lambda: _hip.attention_fwd(q, k, v, 0.125))
    fl = 4.0 * B * H * N * N * 64
    print(f"N = {N}: forward {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s   ({t / (B * H / 256):.2f} us per (batch, head) pair and CU)")
