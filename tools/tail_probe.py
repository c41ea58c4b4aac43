#!/usr/bin/env python3
"""Does a row-streaming kernel pay for a partly filled last round of workgroups?  The plain encoder GEMM (lin_rows_kernel, 128 rows per
workgroup, two workgroups per CU = 512 resident) timed at row counts around whole rounds: if time follows ceil(blocks / 512) the tail
costs a full round; if it follows blocks it does not.   python tools/tail_probe.py [N K]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 256)
w = (torch.randn(N, K, device="cuda:0") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda:0").to(torch.bfloat16)
for blocks in (1024, 1280, 1536, 1540, 1570, 1604, 1664, 1792, 2048):
    M = blocks * 128
    x = torch.randn(M, K, device="cuda:0").to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda:0", dtype=torch.bfloat16)
    for _ in range(3):
        _hip.linear_bf16(x, w, b, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        _hip.linear_bf16(x, w, b, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"N={N} K={K} blocks {blocks:5d} = {blocks / 512:5.2f} rounds: {us:7.1f} us   {us / blocks * 1e3:6.1f} ns per block   {M * (K + N) * 2 / us / 1e6:5.2f} TB/s")
