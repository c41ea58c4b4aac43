import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from viforsdes_amd import _hip
from viforsdes_amd.primitives import fused
dev, M, C, hreal, H = "cuda:0", 4096, 256, 682, 704
g = torch.Generator().manual_seed(0)
P = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(*s, generator=g) * sc).to(dev))
w_in, b_in, w_out, b_out = P(2 * hreal, C, sc=C ** -0.5), P(2 * hreal), P(C, hreal, sc=hreal ** -0.5), P(C)
x = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16)
dy = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16)
pin, pout = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=True)
img = fused.MlpBwdImages(pin, pout, H).operand()
u, _ = _hip.linear_swiglu_bf16(x, *pin.operands(), want_u=True)
du, dx = _hip.mlp_bwd(dy, u, img, H)
du_old = _hip.linear_swiglu_bwd_bf16(dy, pout.transposed(), u)
err = (du.float() - du_old.float()).abs()
tile_err = err.reshape(M, H // 32, 64).amax(dim=(0, 2))
print("max err per pair tile:", [f"{v:.2f}" for v in tile_err.tolist()])
row_err = err.reshape(M // 128, 128, -1).amax(dim=(1, 2))
print("max err per 128-row workgroup (first 16):", [f"{v:.2f}" for v in row_err[:16].tolist()])
bad = err.reshape(M, H // 32, 64)[:128]
t_bad = int(tile_err.argmax())
print("worst tile", t_bad, "rows with error in wg 0:", (bad[:, t_bad].amax(dim=1) > 0.05).nonzero().flatten().tolist()[:40])
print("cols with error in worst tile:", (err.reshape(M, H // 32, 64)[:, t_bad].amax(dim=0) > 0.05).nonzero().flatten().tolist())
