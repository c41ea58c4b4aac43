"""Debug aid: build a VSDE_TRACE variant of the library, run the OU-size forward once and print the
cycle stamps of one wave for one time step (s_memtime), to see where a step's latency goes."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libvsde_trace.so")   # built here or shipped prebuilt (hipcc -DVSDE_TRACE)
sys.path.insert(0, ROOT)
from viforsdes_amd.build import SOURCES
src = [os.path.join(ROOT, "viforsdes_amd/csrc", f) for f in SOURCES]
if not os.path.exists(so): subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DVSDE_TRACE", "-shared", "-fPIC", "-o", so] + src, check=True)
import viforsdes_amd.build as b
b.LIB_PATH = so
from viforsdes_amd import _hip
_hip.LIB_PATH = so
dev = torch.device("cuda:0")
B, T, S, C, P, H, L = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1, 256, 3, 64, 2
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
NO = S + S * (S + 1) // 2
ws = [rn(3*H, S+C+P, sc=.08), rn(3*H, H, sc=.12), rn(3*H, sc=.1), rn(3*H, sc=.1), rn(L-1, 3*H, H, sc=.12), rn(L-1, 3*H, H, sc=.12),
      rn(L-1, 3*H, sc=.1), rn(L-1, 3*H, sc=.1), rn(NO, H, sc=.1), torch.ones(NO).to(dev)]
x0, ctx, theta, eps = rn(B, S), rn(B, T+1, C).to(torch.bfloat16)[:, :-1], rn(B, P).abs(), rn(B, T, S)
for save in (False, True):
    _hip.head_forward(x0, ctx, theta, eps, ws, 0.1, save)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    _hip.load().vsde_debug_read_trace(buf)
    st = list(buf)[:13]
    print("save" if save else "eval", "deltas:", [st[i+1]-st[i] for i in range(12)], "total", st[12]-st[0])
