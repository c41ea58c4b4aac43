#!/usr/bin/env python3
"""Which torch operators launch the small kernels of one LV training step?  torch.profiler (CPU + device activity) over one eager
step: every operator that is not one of the library's own kernels, by self device time.    python tools/step_small_ops.py [lv|ou]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from torch.profiler import profile, ProfilerActivity
from viforsdes_amd.examples.sdes import lv_problem, ou_problem

which = sys.argv[1] if len(sys.argv) > 1 else "lv"
dev = torch.device("cuda:0")
tr = bench.build_trainer(lv_problem() if which == "lv" else ou_problem(), 512 if which == "lv" else 128, dev, True, seed=1234)
model = tr.ctx.model
for _ in range(5):
    tr._train_step(model)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr._train_step(model)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages():
    t = getattr(ev, "self_device_time_total", None)
    if t is None:
        t = getattr(ev, "self_cuda_time_total", 0.0)
    if t > 0:
        rows.append((t, ev.count, ev.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{which}: device time by operator (self), total {tot / 1e3:.2f} ms")
small = 0.0
for t, c, k in rows:
    own = any(s in k for s in ("vsde", "Cijk", "attn_", "lin_rows", "wgrad", "mlp_", "head_", "ln_mod", "tn_wide", "proj_", "elbo_", "optim_", "pack_refresh", "colsum", "gated_residual", "em_"))
    if not own and not k.startswith("void ") and not k.startswith("__amd"):
        small += t
    print(f"{t:9.1f} us  x{c:4d}  {k[:110]}")
print(f"operators other than the library's own launches: {small / 1e3:.2f} ms")

# ---- where they come from: the innermost frame inside the package for every small aten operator (second profiled step, with stacks)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    tr._train_step(model)
    torch.cuda.synchronize()
sites = {}
for ev in prof2.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    dt = sum(getattr(k, "duration", 0.0) for k in ev.kernels) if ev.kernels else 0.0
    if dt <= 0 or dt > 60:
        continue
    frame = next((f for f in ev.stack if "viforsdes_amd" in f and "fused.py" not in f), None) or next((f for f in ev.stack if "viforsdes_amd" in f), None) or (ev.stack[0] if ev.stack else "?")
    key = (frame.split("viforsdes_amd/")[-1][:90], ev.name)
    s = sites.setdefault(key, [0, 0.0])
    s[0] += 1; s[1] += dt
print("small aten launches by source line (count, us):")
for (frame, name), (c, t) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t:7.1f} us x{c:3d}  {name:28s} {frame}")
print(f"sum {sum(v[1] for v in sites.values()) / 1e3:.2f} ms in {sum(v[0] for v in sites.values())} launches")
