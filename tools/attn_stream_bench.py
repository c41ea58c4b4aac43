"""Streamed attention kernels at the config-5 shape (B=256, N=1001, H=4, head_dim 128; AS_B / AS_N / AS_H / AS_D override):
forward and backward times."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip
B, N, H, D = (int(os.environ.get("AS_" + k, v)) for k, v in (("B", 256), ("N", 1001), ("H", 4), ("D", 128)))
g = torch.Generator().manual_seed(0)
q, k, v, go = (torch.randn(B, N, H, D, generator=g).to("cuda:0", torch.bfloat16) for _ in range(4))
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
sc = D ** -0.5
o, lse = _hip.attention_fwd(q, k, v, sc)
print(f"forward  {t(lambda: _hip.attention_fwd(q, k, v, sc)):8.1f} us")
print(f"backward {t(lambda: _hip.attention_bwd(go, q, k, v, o, lse, sc)):8.1f} us")
