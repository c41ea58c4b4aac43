"""Times the kernels of the attention block's training core at the LV bench dims (B=512, N=401, H=4, d=64), next to the
separate-pass kernels they replace.  usage: python tools/attn_core_bench.py [B N]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from viforsdes_amd import _hip

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 401)
H, dev = 4, "cuda:0"
M = B * N
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16)
x, w, b = R(M, 256), R(832, 256) * 0.06, R(832) * 0.1
cos, sin = torch.cos(torch.rand(N, 32, generator=g) * 6).to(dev), torch.sin(torch.rand(N, 32, generator=g) * 6).to(dev)
wq = torch.ones(64, device=dev); wk = torch.ones(64, device=dev)
v0 = R(M, 256); lam = torch.tensor([0.4], device=dev)
dout = R(B, N, H, 64)
dy = torch.empty(M, 832, device=dev, dtype=torch.bfloat16)
acc = torch.zeros(B, N, H, 64, device=dev, dtype=torch.bfloat16)


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


q, k, v, glog, rinv, vdiff = _hip.linear_qknorm_bf16(x, w, b, H, N, cos, sin, wq, wk, v0, lam, 1e-6, save=True)
sh = (B, N, H, 64)
q, k, v = q.view(sh), k.view(sh), v.view(sh)
og, lse = _hip.attention_fwd_gated(q, k, v, glog, 0.125)   # glog: sigmoid factors (save=True)
dattn, delta = _hip.gate_bwd_delta(dout, og, glog, dy[:, 768:])
print(f"projection + qknorm epilogue (save)   {t(lambda: _hip.linear_qknorm_bf16(x, w, b, H, N, cos, sin, wq, wk, v0, lam, 1e-6, save=True)):8.1f} us")
print(f"projection + qknorm epilogue (nograd) {t(lambda: _hip.linear_qknorm_bf16(x, w, b, H, N, cos, sin, wq, wk, v0, lam, 1e-6)):8.1f} us")
print(f"plain projection N=832                {t(lambda: _hip.linear_bf16(x, w, b)):8.1f} us")
y = _hip.linear_bf16(x, w, b).view(B, N, 832)
print(f"qk_norm_rope_fwd                      {t(lambda: _hip.qk_norm_rope_fwd(y[..., :768], cos, sin, wq, wk, v0.view(sh), lam, H, 1e-6, True)):8.1f} us")
print(f"attention fwd gated                   {t(lambda: _hip.attention_fwd_gated(q, k, v, glog, 0.125)):8.1f} us")
print(f"attention fwd                         {t(lambda: _hip.attention_fwd(q, k, v, 0.125)):8.1f} us")
print(f"gate_bwd_delta                        {t(lambda: _hip.gate_bwd_delta(dout, og, glog, dy[:, 768:])):8.1f} us")
print(f"attention bwd fused (mix, accumulate) {t(lambda: _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, vdiff, lam, acc, None, dy, 0.125)):8.1f} us")
print(f"attention bwd fused (no mix, extra)   {t(lambda: _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, None, None, None, acc, dy, 0.125)):8.1f} us")
print(f"attention bwd fused (no mix)          {t(lambda: _hip.attention_bwd_fused(dattn, q, k, v, lse, delta, rinv, cos, sin, wq, wk, None, None, None, None, dy, 0.125)):8.1f} us")
o, _ = _hip.attention_fwd(q, k, v, 0.125)
print(f"attention bwd (separate)              {t(lambda: _hip.attention_bwd(dattn, q, k, v, o, lse, 0.125)):8.1f} us")
dq, dk, dv = _hip.attention_bwd(dattn, q, k, v, o, lse, 0.125)
print(f"qk_norm_rope_bwd                      {t(lambda: _hip.qk_norm_rope_bwd(y[..., :768], cos, sin, wq, wk, v0.view(sh), lam, dq, dk, dv, H, 1e-6, True, dqkv=dy.view(B, N, 832)[..., :768], dv0=acc)):8.1f} us")
wo = R(256, 256) * 0.06
dy_o = R(M, 256)
print(f"out-proj dgrad + gate backward (1 GEMM) {t(lambda: _hip.linear_gate_bwd(dy_o, wo, og, glog, dy[:, 768:], N)):8.1f} us")
print(f"out-proj dgrad (plain)                {t(lambda: _hip.linear_bf16(dy_o, wo, None)):8.1f} us")
