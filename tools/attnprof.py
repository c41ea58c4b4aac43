"""Launch the attention kernels a few times at the LV shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from viforsdes_amd import _hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(512, 401, 4, 64, generator=g).to(dev, torch.bfloat16) for _ in range(3))
go = torch.randn_like(q)
for _ in range(10):
    o, lse = _hip.attention_fwd(q, k, v, 0.125)
    _hip.attention_bwd(go, q, k, v, o, lse, 0.125)
torch.cuda.synchronize()
