#!/usr/bin/env python3
"""Launches the fused SwiGLU MLP kernels (csrc/vsde_mlp.hip) a few times at the LV shape -- the driver for tools/pmc.sh / rocprofv3:
    tools/pmc.sh mlp_ tools/mlp_probe.py [fwd|train|bwd] [M]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402
from viforsdes_amd.primitives import fused  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 205312
dev, C, hreal, H = "cuda:0", 256, 682, 704
P = lambda *s, sc=1.0: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)
w_in, b_in, w_out, b_out = P(2 * hreal, C, sc=C ** -0.5), P(2 * hreal), P(C, hreal, sc=hreal ** -0.5), P(C)
x = torch.randn(M, C, device=dev).to(torch.bfloat16)
pin, pout = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=False)
img = fused.MlpImages(pin, pout, H)
w1, w2, b1 = img.operands()
if mode == "bwd":
    pin_i, pout_i = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=True)
    bimg = fused.MlpBwdImages(pin_i, pout_i, H).operand()
    u, _ = _hip.linear_swiglu_bf16(x, *pin_i.operands(), want_u=True)
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
for _ in range(5):
    if mode in ("fwd", "train"):
        _hip.mlp_fwd(x, w1, w2, b1, pout.bias, H, want_s=mode == "train")
    elif mode == "bwd":
        _hip.mlp_bwd(dy, u, bimg, H)
torch.cuda.synchronize()
