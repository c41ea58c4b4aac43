"""SwiGLU input projection (lin_rows_kernel<256, EPI_SWIGLU>) against the number of row stripes: is the 1.57-round grid of the LV
shape (802 workgroups on 512 slots) paying for two full rounds?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip
K, N = 256, 1536
dev = "cuda:0"
w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16); b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (65536, 131072, 163840, 196608, 205312, 229376, 262144, 393216):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: _hip.linear_swiglu_bf16(x, w, b)); t0 = timeit(lambda: _hip.linear_swiglu_bf16(x, w, b, want_u=False))
    print(f"M={M:7d} ({M / 256 / 512:4.2f} rounds of 512 workgroups): with u {t1:6.1f} us = {t1 / M * 1e3:5.3f} ns/row | without u {t0:6.1f} us = {t0 / M * 1e3:5.3f} ns/row")
