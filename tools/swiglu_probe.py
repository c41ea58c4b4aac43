import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from viforsdes_amd import _hip
M, K, N = 205312, 256, 1536
dev = "cuda:0"
x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16); b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("SwiGLU-in with u: %.1f us | without u: %.1f us" % (timeit(lambda: _hip.linear_swiglu_bf16(x, w, b)), timeit(lambda: _hip.linear_swiglu_bf16(x, w, b, want_u=False))))
