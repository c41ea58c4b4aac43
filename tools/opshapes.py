#!/usr/bin/env python3
"""Small torch kernels (<= 60 us) of one training step by aten op AND input shapes: calls per step, device us per step.
    python tools/opshapes.py [lv|ou]"""
import os, sys, collections, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.getcwd())
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem, ou_problem
wl = sys.argv[1] if len(sys.argv) > 1 else "lv"
tr = build_trainer(lv_problem() if wl == "lv" else ou_problem(), 512 if wl == "lv" else 128, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(3):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        tr._train_step(tr.ctx.model); tr.ctx.ema.update()
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.self_device_time_total > 0 and e.self_device_time_total <= 60 and e.name.startswith("aten::"):
        key = (e.name, str(e.input_shapes)[:110])
        by[key][0] += 1; by[key][1] += e.self_device_time_total
for (n, sh), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{c/2:6.1f} {t/2:8.1f} us  {n:28s} {sh}")
