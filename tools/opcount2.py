#!/usr/bin/env python3
"""Which torch ops launch the small kernels of one LV training step?  torch.profiler over 2 eager steps: per aten op the number of
calls per step and the device time, sorted by device time (own HIP kernels launched through ctypes do not appear as ops)."""
import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem
tr = build_trainer(lv_problem(), 512, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(3):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(2):
        tr._train_step(tr.ctx.model); tr.ctx.ema.update()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
print(f"{'op':40s} {'calls/step':>10s} {'self dev us/step':>16s}  shapes")
for e in rows[:60]:
    if e.self_device_time_total <= 0:
        continue
    print(f"{e.key[:40]:40s} {e.count / 2:10.1f} {e.self_device_time_total / 2:16.1f}  {str(e.input_shapes)[:110]}")
