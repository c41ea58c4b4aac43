import os as _os; _os.environ.setdefault("VSDE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "viforsdes_amd", "libvsde_hip_abl.so"))  # the tools' library: A/B switches + variants (python -m viforsdes_amd.build --ablations)
import os, sys, torch
sys.path.insert(0, os.getcwd())
from viforsdes_amd import _hip
B, N, H = 512, 401, 4
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to("cuda:0", torch.bfloat16)
q, k, v = R(B, N, H, 64), R(B, N, H, 64), R(B, N, H, 64)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get("VSDE_ATTN_FWD_ABL", "0"), f"attention fwd {t(lambda: _hip.attention_fwd(q, k, v, 0.125)):8.1f} us")
