"""Time the fused norm / residual backward kernels at the LV encoder shape (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from viforsdes_amd import _hip

B, N, C = 512, 401, 256
dev = torch.device("cuda:0")
bf = torch.bfloat16
x = torch.randn(B, N, C, device=dev, dtype=bf); dy = torch.randn_like(x); dres = torch.randn_like(x); y = torch.randn_like(x)
sc = torch.randn(B, C, device=dev, dtype=bf); sh = torch.randn_like(sc); gate = torch.randn_like(sc)
_, mean, rstd = _hip.ln_modulate_fwd(x, sc, sh, 1e-5)

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

mb = B * N * C * 2 / 1e6
print(f"ln_fwd            {t(lambda: _hip.ln_modulate_fwd(x, sc, sh, 1e-5)):7.1f} us  ({2*mb:.0f} MB)")
print(f"ln_bwd (+dres)    {t(lambda: _hip.ln_modulate_bwd(x, sc, dy, mean, rstd, dres)):7.1f} us  ({4*mb:.0f} MB)")
print(f"ln_bwd            {t(lambda: _hip.ln_modulate_bwd(x, sc, dy, mean, rstd)):7.1f} us  ({3*mb:.0f} MB)")
print(f"gres_fwd          {t(lambda: _hip.gated_residual_fwd(x, y, gate)):7.1f} us  ({3*mb:.0f} MB)")
print(f"gres_bwd          {t(lambda: _hip.gated_residual_bwd(y, gate, dy)):7.1f} us  ({3*mb:.0f} MB)")
xn, h, *_ = _hip.residual_ln_fwd(x, y, gate, sc, sh, 1e-5) if hasattr(_hip, "residual_ln_fwd") else (None, None)
if hasattr(_hip, "residual_ln_fwd"):
    print(f"residual_ln_fwd   {t(lambda: _hip.residual_ln_fwd(x, y, gate, sc, sh, 1e-5)):7.1f} us  ({4*mb:.0f} MB)")
