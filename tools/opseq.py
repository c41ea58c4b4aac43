#!/usr/bin/env python3
"""The device kernels of ONE training step in launch order (name, duration, the aten op that launched it with its input shapes):
    python tools/opseq.py [lv|ou] [max_us]      kernels longer than max_us (default 60) are printed as '...' separators"""
import os, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.getcwd())
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem, ou_problem
wl = sys.argv[1] if len(sys.argv) > 1 else "lv"
cap = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
tr = build_trainer(lv_problem() if wl == "lv" else ou_problem(), 512 if wl == "lv" else 128, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(3):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
    torch.cuda.synchronize()
ops = sorted((e for e in prof.events() if e.device_type.name == "CPU" and e.name.startswith("aten::") and e.self_device_time_total > 0),
             key=lambda e: e.time_range.start)
big = 0
for e in ops:
    if e.self_device_time_total > cap:
        big += 1
        continue
    if big:
        print(f"      ... {big} larger kernels ...")
        big = 0
    print(f"{e.self_device_time_total:7.1f} us  {e.name:30s} {str(e.input_shapes)[:120]}")
