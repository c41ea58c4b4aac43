"""Run PyTorch TunableOp over the encoder's GEMM shapes (LV) and write the selected solutions to a CSV (GPU only).

    python tools/tunegemm.py out.csv
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F

out = sys.argv[1] if len(sys.argv) > 1 else "tunableop_results.csv"
dev = torch.device("cuda:0")
M = 512 * 401
bf = torch.bfloat16
shapes = [  # (name, kind, N, K): fwd = F.linear(x[M,K], W[N,K], b) ; dgrad = dy[M,N] @ W[N,K]
    ("proj", 832, 256), ("out", 256, 256), ("mlp_in", 1536, 256), ("mlp_out", 256, 768),
    ("mlp_in704", 1408, 256), ("mlp_out704", 256, 704)]   # SwiGLU width 682 padded to 704 (round 3)
if len(sys.argv) > 2:   # only the named shapes
    shapes = [s_ for s_ in shapes if s_[0] in sys.argv[2:]]

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

ops = []
for name, N, K in shapes:
    x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf) * 0.05
    b = torch.randn(N, device=dev, dtype=bf); dy = torch.randn(M, N, device=dev, dtype=bf)
    ops.append((name + " fwd", lambda x=x, w=w, b=b: F.linear(x, w, b)))
    ops.append((name + " dgrad", lambda dy=dy, w=w: dy @ w))
base = {n: bench(f) for n, f in ops}
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True); tn.set_filename(out)
tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(50)
t0 = time.time()
for n, f in ops:
    f(); torch.cuda.synchronize()
print(f"tuning took {time.time() - t0:.1f} s")
tuned = {n: bench(f) for n, f in ops}
(tn.write_file(out) if hasattr(tn, "write_file") else None)
for n, _ in ops:
    print(f"{n:14s} default {base[n]:7.1f} us   tuned {tuned[n]:7.1f} us   ({base[n]/tuned[n]:.2f}x)")
print(f"sum default {sum(base.values()):.0f} us  tuned {sum(tuned.values()):.0f} us")
