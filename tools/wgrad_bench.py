#!/usr/bin/env python3
"""Times vsde_linear_wgrad_bf16 on the LV encoder's four Linear shapes against its HBM floor (dy and x read once)."""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402

M = 205312
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in (("qkv+gate", 832, 256), ("out_proj", 256, 256), ("mlp.in", 1408, 256), ("mlp.out", 256, 704)):
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    t = timeit(lambda: _hip.linear_wgrad(dy, x, True))
    by = 2.0 * M * (N + K)
    print(f"{name:10s} dW[{N},{K}]: {t:7.1f} us  = {by / t / 1e6:5.2f} TB/s of operand bytes, {2.0 * M * N * K / t / 1e6:6.0f} TF/s")
