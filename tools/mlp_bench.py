#!/usr/bin/env python3
"""Times the fused SwiGLU MLP kernels (csrc/vsde_mlp.hip) against the two-launch chain they replace, at the LV encoder's shape.
    python tools/mlp_bench.py [--m 205312]"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip  # noqa: E402
from viforsdes_amd.accelerate import enable_tuned_gemms  # noqa: E402
from viforsdes_amd.primitives import fused  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=205312)
ap.add_argument("--c", type=int, default=256)
ap.add_argument("--h", type=int, default=682)
a = ap.parse_args()
enable_tuned_gemms()
dev = "cuda:0"
M, C, hreal = a.m, a.c, a.h
H = -(-hreal // 64) * 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


P = lambda *s, sc=1.0: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)
w_in, b_in, w_out, b_out = P(2 * hreal, C, sc=C ** -0.5), P(2 * hreal), P(C, hreal, sc=hreal ** -0.5), P(C)
x = torch.randn(M, C, device=dev).to(torch.bfloat16)
pin_i, pout_i = fused.swiglu_packs(w_in, b_in, w_out, b_out, H, interleave=True)
img = fused.MlpImages(pin_i, pout_i, H)
w1, w2, b1 = img.operands()
w1i, b1i = pin_i.operands(); w2o, b2o = pout_i.operands()
print(f"M = {M}, C = {C}, H = {H} ({hreal})")
fl = 2.0 * M * (2 * H * C + H * C)
for name, fn, by in (
        ("old no-grad: rows+SwiGLU | hipBLASLt", lambda: torch.nn.functional.linear(_hip.linear_swiglu_bf16(x, w1i, b1i, want_u=False)[1], w2o, b2o), 0),
        ("old train  : rows+SwiGLU | hipBLASLt", lambda: torch.nn.functional.linear(_hip.linear_swiglu_bf16(x, w1i, b1i, want_u=True)[1], w2o, b2o), 0),
        ("fused no-grad", lambda: _hip.mlp_fwd(x, w1, w2, b1, b2o, H, want_s=False), 2.0 * M * 2 * C),
        ("fused train (writes s)", lambda: _hip.mlp_fwd(x, w1, w2, b1, b2o, H, want_s=True), 2.0 * M * (2 * C + H))):
    t = timeit(fn)
    print(f"{name:40s} {t:8.1f} us   {fl / t / 1e6:6.0f} TF/s" + (f"   {by / t / 1e3:6.0f} GB/s algorithmic" if by else ""))
# backward: the rows kernel with the SwiGLU derivative + the library GEMM over du, against the fused kernel
dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
u, _ = _hip.linear_swiglu_bf16(x, w1i, b1i, want_u=True)
w2t, w1p = pout_i.transposed(), pin_i.weight
bimg = fused.MlpBwdImages(pin_i, pout_i, H).operand()
flb = 2.0 * M * (H * C + 2 * H * C)
for name, fn in (("old bwd: rows+SwiGLU' | hipBLASLt dx", lambda: _hip.linear_swiglu_bwd_bf16(dy, w2t, u) @ w1p),
                 ("fused bwd (du + dx)", lambda: _hip.mlp_bwd(dy, u, bimg, H))):
    t = timeit(fn)
    print(f"{name:40s} {t:8.1f} us   {flb / t / 1e6:6.0f} TF/s   {2.0 * M * (2 * C + 4 * H) / t / 1e3:6.0f} GB/s algorithmic")
# block forms (no-grad sampling): [out projection |] residual + LN | MLP | residual + next LN
if M % 401 == 0 and C in (128, 256):
    B, N = M // 401, 401
    xs, attn = torch.randn(B, N, C, device=dev).to(torch.bfloat16), torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    glog = torch.randn(M, 64, device=dev).to(torch.bfloat16)
    allm = (0.5 * torch.randn(B, 6 * C, device=dev)).to(torch.bfloat16)
    ga, sc, sh, gm, sn, hs = [allm[:, i * C:(i + 1) * C] for i in range(6)]
    po = fused.plain_pack(P(C, C, sc=C ** -0.5), P(C))
    wo, bo = po.operands()
    oimg = fused.OutProjImage(po).operand()
    for name, fn in (
            ("gated out projection (rows kernel)", lambda: _hip.linear_gated_bf16(attn.view(M, C), glog, wo, bo)),
            ("block form", lambda: _hip.mlp_block_fwd(xs, attn, ga, sc, sh, gm, sn, hs, 1e-5, 1e-5, w1, w2, b1, b2o, H)),
            ("block form + out projection", lambda: _hip.mlp_attn_block_fwd(xs, attn, glog, oimg, bo, ga, sc, sh, gm, sn, hs, 1e-5, 1e-5, w1, w2, b1, b2o, H))):
        print(f"{name:40s} {timeit(fn):8.1f} us")

# the deep-reduction GEMMs at width 256: library (torch.nn.functional.linear / matmul) against csrc/vsde_mlp.hip::deep256_kernel
for name, K, tr in (("mlp out projection  [M,704] x [256,704]^T", 704, False), ("mlp input gradient  [M,1408] x [1408,256]", 1408, True),
                    ("qkv input gradient  [M,832] x [832,256]", 832, True)):
    wt = P(*((K, 256) if tr else (256, K)), sc=K ** -0.5)
    pk = fused.plain_pack(wt, None)
    xx = torch.randn(M, K, device=dev).to(torch.bfloat16)
    wb = pk.weight
    t_lib = timeit(lambda: (xx @ wb) if tr else torch.nn.functional.linear(xx, wb))
    t_own = timeit(lambda: fused.deep256(xx, pk, tr, None))
    print(f"{name:46s} library {t_lib:7.1f} us   own {t_own:7.1f} us   ({2.0 * M * K * 256 / t_own / 1e6:5.0f} TF/s)")
