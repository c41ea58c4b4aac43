#!/usr/bin/env python3
"""Ablation of the persistent deep-reduction GEMM: VSDE_LIN_DEBUG = 0 full, 2 no activation DMA, 3 no weight DMA, 4 no DMA, 5 no MFMA."""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
from viforsdes_amd import _hip
M = 205312
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for N, K in ((256, 704), (256, 768), (256, 1408)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    out.append(f"K={K}: {timeit(lambda: _hip.linear_bf16(x, w, None)):.1f} us")
print("VSDE_LIN_DEBUG=" + os.environ.get("VSDE_LIN_DEBUG", "0"), " | ".join(out))
'''
for dbg in ("0", "2", "3", "4", "5", "6", "7"):   # 6: activation DMA only, 7: weight DMA only (no MFMAs)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VSDE_LIN_DEBUG=dbg, VSDE_DEEP_GEMM="1"))
