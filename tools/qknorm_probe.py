"""Time of the no-grad projection kernel (vsde_linear_qknorm_bf16) against GEMM + qk_norm_rope at the LV shape."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip
B, N, K, heads = 512, 401, 256, 4
M = B * N
dev = "cuda:0"
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(832, K, device=dev) * 0.05).to(torch.bfloat16); b = torch.zeros(832, device=dev, dtype=torch.bfloat16)
cos = torch.rand(N, 32, device=dev); sin = torch.rand(N, 32, device=dev); wq = torch.ones(64, device=dev); wk = torch.ones(64, device=dev)
v0 = torch.randn(M, 256, device=dev).to(torch.bfloat16); lam = torch.tensor([0.5], device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def unfused():
    y = _hip.linear_bf16(x, w, b)
    return _hip.qk_norm_rope_fwd(y.view(B, N, 832)[..., :768], cos, sin, wq, wk, v0.view(B, N, heads, 64), lam, heads, 1e-6, True)
print("GEMM + qk_norm_rope: %.1f us | fused epilogue (with v0): %.1f us | fused (no v0): %.1f us" % (
    timeit(unfused), timeit(lambda: _hip.linear_qknorm_bf16(x, w, b, heads, N, cos, sin, wq, wk, v0, lam, 1e-6)),
    timeit(lambda: _hip.linear_qknorm_bf16(x, w, b, heads, N, cos, sin, wq, wk, None, None, 1e-6))))
