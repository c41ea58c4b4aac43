#!/usr/bin/env python3
"""Small torch kernels of one LV training step by aten op (all shapes merged): calls per step, device us per step; and the same
grouped by the python frame that issued them (with_stack)."""
import os, sys, collections
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem, ou_problem
wl = sys.argv[1] if len(sys.argv) > 1 else "lv"
tr = build_trainer(lv_problem() if wl == "lv" else ou_problem(), 512 if wl == "lv" else 128, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(3):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=True) as prof:
    for _ in range(2):
        tr._train_step(tr.ctx.model); tr.ctx.ema.update()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_time_total > 0 and e.name.startswith("aten::")]
byop = collections.defaultdict(lambda: [0, 0.0])
byframe = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.self_device_time_total <= 0 or e.self_device_time_total > 60:
        continue
    byop[e.name][0] += 1; byop[e.name][1] += e.self_device_time_total
    frame = next((s for s in (e.stack or []) if ("viforsdes_amd/" in s or "bench.py" in s) and "torch/" not in s), "?")
    frame = frame.split("viforsdes_amd/")[-1][:110] if "viforsdes_amd/" in frame else frame[-110:]
    byframe[frame][0] += 1; byframe[frame][1] += e.self_device_time_total
print("== by op (kernels <= 60 us) ==")
for k, (n, t) in sorted(byop.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:45s} {n / 2:7.1f} calls/step {t / 2:8.1f} us/step")
print("total:", sum(v[0] for v in byop.values()) / 2, "calls/step,", sum(v[1] for v in byop.values()) / 2, "us/step")
if all(k == "?" for k in byframe):
    print("no python frames in the events; sample stack:", next((e.stack for e in ev if e.stack), None))
print("== by frame ==")
for k, (n, t) in sorted(byframe.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n / 2:6.1f} {t / 2:8.1f} us  {k}")
