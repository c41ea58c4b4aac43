#!/usr/bin/env python3
"""Times the encoder GEMM kernels (csrc/vsde_linear.hip) against torch.nn.functional.linear (hipBLASLt) on the LV shapes.
    python tools/linear_bench.py [--m 205312]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402
from viforsdes_amd.accelerate import enable_tuned_gemms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=205312)
a = ap.parse_args()
enable_tuned_gemms()
dev = "cuda:0"
M = a.m


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"M = {M}")
for name, N, K in (("qkv+gate fwd", 832, 256), ("out_proj fwd/dgrad", 256, 256), ("mlp.out fwd", 256, 768),
                   ("qkv dgrad", 256, 832), ("mlp.in dgrad", 256, 1536), ("mlp.in fwd (plain)", 1536, 256), ("mlp.out dgrad (plain)", 768, 256),
                   ("mlp.out fwd 704", 256, 704), ("mlp.in dgrad 1408", 256, 1408)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_own = timeit(lambda: _hip.linear_bf16(x, w, b))
    fl, by = 2.0 * M * N * K, 2.0 * (M * K + M * N)
    print(f"{name:24s} N={N:5d} K={K:5d}: hipBLASLt {t_lib:7.1f} us | own {t_own:7.1f} us = {fl / t_own / 1e6:6.0f} TF/s, {by / t_own / 1e3:6.0f} GB/s")
# fused SwiGLU pair vs the unfused chain
K, H = 256, 704   # the padded SwiGLU width of the LV encoder (682 -> 704)
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w1 = (torch.randn(2 * H, K, device=dev) * K ** -0.5).to(torch.bfloat16); b1 = torch.randn(2 * H, device=dev).to(torch.bfloat16)
w2t = (torch.randn(H, K, device=dev) * H ** -0.5).to(torch.bfloat16)
dy = torch.randn(M, K, device=dev).to(torch.bfloat16)
u = torch.nn.functional.linear(x, w1, b1)
t_f = timeit(lambda: _hip.swiglu_fwd(torch.nn.functional.linear(x, w1, b1)))
t_fo = timeit(lambda: _hip.linear_swiglu_bf16(x, w1, b1))
t_b = timeit(lambda: _hip.swiglu_bwd(u, dy @ w2t.t()))
t_bo = timeit(lambda: _hip.linear_swiglu_bwd_bf16(dy, w2t, u))
print(f"mlp.in + SwiGLU fwd: hipBLASLt + swiglu kernel {t_f:7.1f} us | fused {t_fo:7.1f} us")
print(f"mlp.out dgrad + SwiGLU bwd: hipBLASLt + swiglu_bwd kernel {t_b:7.1f} us | fused {t_bo:7.1f} us")
# config-5 shapes (encoder 512, M = 256 x 1001)
M5 = 256 * 1001
print(f"config 5: M = {M5}")
for name, N, K in (("qkv+gate fwd", 1664, 512), ("out_proj fwd/dgrad", 512, 512), ("mlp.out fwd", 512, 1408), ("qkv dgrad", 512, 1664),
                   ("mlp.in dgrad", 512, 2816)):
    x = torch.randn(M5, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
    t_own = timeit(lambda: _hip.linear_bf16(x, w, b))
    fl, by = 2.0 * M5 * N * K, 2.0 * (M5 * K + M5 * N)
    print(f"{name:24s} N={N:5d} K={K:5d}: hipBLASLt {t_lib:7.1f} us | own {t_own:7.1f} us = {fl / t_own / 1e6:6.0f} TF/s, {by / t_own / 1e3:6.0f} GB/s")
