#!/usr/bin/env python3
"""Where do the full-size ([512,401,256]) torch ops of an LV training step come from?  TorchDispatchMode prints the python stack of
every aten fill / zeros / sum / add / copy on a tensor of that size."""
import os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem
tr = build_trainer(lv_problem(), 512, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(2):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
BIG = 512 * 401 * 256
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        ts = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor)]
        if any(t.numel() >= BIG for t in ts) and any(k in name for k in ("fill", "zero", "sum", "add", "copy", "clone", "contiguous", "mul", "_to_copy")):
            frames = [f for f in traceback.extract_stack() if "/root/repo" in f.filename or "viforsdes_amd" in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-4:][::-1])
            print(f"{name:38s} {[tuple(t.shape) for t in ts][:3]}  {where}")
        return out
with Spy():
    tr._train_step(tr.ctx.model)
torch.cuda.synchronize()
