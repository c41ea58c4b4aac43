import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from viforsdes_amd import _hip
M, K, H = 205312, 256, 704
dev = "cuda:0"
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w1 = (torch.randn(2 * H, K, device=dev) * K ** -0.5).to(torch.bfloat16); b1 = torch.randn(2 * H, device=dev).to(torch.bfloat16)
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("eval %.1f us  train %.1f us" % (timeit(lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=False)), timeit(lambda: _hip.linear_swiglu_bf16(x, w1, b1, want_u=True))))
