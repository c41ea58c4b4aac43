#!/usr/bin/env python3
"""The three deep-reduction GEMMs of a SiT block at the LV shape (M = 205,312 rows, 256 outputs; K = 704 | 1408 | 832): hipBLASLt against
csrc/vsde_mlp.hip::deep256q_kernel (+ the 64-row tail launch), alternating launches on one box.    python tools/deep_bench.py [M]
(VSDE_DEEP256_Q=0: deep256p_kernel; VSDE_DEEP256_TAIL=0: one launch of 256-row workgroups; VSDE_DEEP256_ABL: timing-only ablations)"""
import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("VSDE_DEEP256", "1")
from viforsdes_amd.primitives import fused

M = int(sys.argv[1]) if len(sys.argv) > 1 else 205312
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"M = {M}  (Q={os.environ.get('VSDE_DEEP256_Q', '1')}, TAIL={os.environ.get('VSDE_DEEP256_TAIL', '1')})")
for name, K, tr in (("mlp out projection  K=704 ", 704, False), ("mlp input gradient  K=1408", 1408, True), ("qkv input gradient  K=832 ", 832, True)):
    w = torch.nn.Parameter((torch.randn(*((K, 256) if tr else (256, K))) * K ** -0.5).to(dev))
    pk = fused.plain_pack(w, None)
    xx = torch.randn(M, K, device=dev).to(torch.bfloat16)
    wb = pk.operands()[0]
    lib = lambda: (xx @ wb) if tr else torch.nn.functional.linear(xx, wb)
    own = lambda: fused.deep256(xx, pk, tr, None)
    ref, got = lib().float(), own().float()
    err = float((got - ref).abs().max() / ref.abs().max())
    ts = [(timeit(lib), timeit(own)) for _ in range(3)]
    tl, to = min(t[0] for t in ts), min(t[1] for t in ts)
    floor = (M * K * 2 + M * 256 * 2) / 5.25e6   # us at 5.25 TB/s
    print(f"{name}  library {tl:7.1f} us   own {to:7.1f} us   ({2.0 * M * K * 256 / to / 1e6:5.0f} TF/s, {(M * K * 2 + M * 512) / to / 1e3:5.0f} GB/s; "
          f"HBM floor ~{floor:5.1f} us)   max err vs library {err:.1e}")
