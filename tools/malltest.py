"""Does a consumer kernel run faster when its input was just produced and still sits in the 256 MB Infinity Cache? (GPU only)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from viforsdes_amd import _hip

dev = torch.device("cuda:0")
for rows in [12800, 25600, 51200, 102400, 204800]:
    u = torch.randn(rows, 1536, device=dev, dtype=torch.bfloat16)
    x = torch.randn(rows, 256, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1536, 256, device=dev, dtype=torch.bfloat16)
    evs = []
    for it in range(12):
        torch.mm(x, w.t(), out=u)            # producer: GEMM writes u
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _hip.swiglu_fwd(u); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[2:])
    t = ts[len(ts) // 2] * 1e3
    mb = rows * 1536 * 2 * 1.5 / 1e6
    print(f"rows {rows:7d}  u {rows*1536*2/1e6:6.0f} MB  swiglu {t:7.1f} us  -> {mb/t*1e3/1e3:6.2f} TB/s   ({t/rows*204800:7.1f} us per 204800 rows)")
