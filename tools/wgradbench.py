"""Time vsde_linear_wgrad_bf16 on the LV encoder's weight-gradient shapes (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from viforsdes_amd import _hip

M = 512 * 401
dev = torch.device("cuda:0")
shapes = [("qkv+gate", 832, 256), ("out", 256, 256), ("mlp_in", 1536, 256), ("mlp_out", 256, 768)]
if os.environ.get("WGRAD_SMALL"):
    shapes = [("gate", 64, 256), ("n128", 128, 256), ("k128", 512, 128)]
tot = 0.0
for name, N, K in shapes:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _hip.linear_wgrad(dy, x, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _hip.linear_wgrad(dy, x, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms
    print(f"{name:9s} N={N:5d} K={K:4d}  {ms*1e3:7.1f} us   min-traffic {(M*(N+K)*2)/ms/1e6:7.1f} GB/s")
print(f"total {tot*1e3:.1f} us per block")
