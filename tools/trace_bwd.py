"""Debug aid: cycle stamps of one wave for one reverse time step of head_bwd_v2_kernel (VSDE_TRACE build)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libvsde_trace.so")   # built here or shipped prebuilt (hipcc -DVSDE_TRACE)
from viforsdes_amd.build import SOURCES
src = [os.path.join(ROOT, "viforsdes_amd/csrc", f) for f in SOURCES]
if not os.path.exists(so): subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DVSDE_TRACE", "-shared", "-fPIC", "-o", so] + src, check=True)
import viforsdes_amd.build as b
b.LIB_PATH = so
from viforsdes_amd import _hip
_hip.LIB_PATH = so
dev = torch.device("cuda:0")
B, T, S, C, P, H, L = int(sys.argv[2]) if len(sys.argv) > 2 else 512, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 2, 256, 3, 64, 2
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
NO = S + S * (S + 1) // 2
ws = [rn(3*H, S+C+P, sc=.08), rn(3*H, H, sc=.12), rn(3*H, sc=.1), rn(3*H, sc=.1), rn(L-1, 3*H, H, sc=.12), rn(L-1, 3*H, H, sc=.12),
      rn(L-1, 3*H, sc=.1), rn(L-1, 3*H, sc=.1), rn(NO, H, sc=.1), torch.ones(NO).to(dev)]
x0, ctx, theta, eps = rn(B, S), rn(B, T+1, C)[:, :-1], rn(B, P).abs(), rn(B, T, S)
gp, gm, gl = rn(B, T+1, S), rn(B, T, S), rn(B, T, S, S)
out = _hip.head_forward(x0, ctx, theta, eps, ws, 0.1, True)
_hip.head_backward(gp, gm, gl, ctx, theta, eps, out[0], out[3], out[4], ws, 0.1)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
_hip.load().vsde_debug_read_trace(buf)
st = list(buf)
names = ["acts+upstream", "dcur (emission^T)", "l1 gates", "l1 barrier", "l1 products", "l0 gates", "l0 barrier", "l0 products"]
idx = [20, 21, 22, 23, 24, 25, 26, 27, 28]
print({n: st[idx[i+1]] - st[idx[i]] for i, n in enumerate(names)}, "step total", st[31] - st[20])
