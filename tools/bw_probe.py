#!/usr/bin/env python3
"""HBM reference rates on this box: fill (write-only), copy (read + write), read-reduce, for ~630 MB bf16 buffers."""
import torch
dev = "cuda:0"
n = 205312 * 1536
a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
a.normal_()
def t(fn, k=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e-3
by = n * 2
print(f"fill  : {by / t(lambda: b.fill_(1.0)) / 1e12:.2f} TB/s (write only)")
print(f"copy  : {2 * by / t(lambda: b.copy_(a)) / 1e12:.2f} TB/s (read + write)")
print(f"sum   : {by / t(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32)) / 1e12:.2f} TB/s (read only)")
