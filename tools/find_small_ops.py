#!/usr/bin/env python3
"""All kernel-launching aten ops of one LV training step by python frame."""
import os, sys, traceback, collections
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer
from viforsdes_amd.examples.sdes import lv_problem
tr = build_trainer(lv_problem(), 512, torch.device("cuda:0"), True, seed=1, enc_hidden=256, enc_depth=8)
for _ in range(2):
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
seen = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(k in name for k in ("view", "reshape", "detach", "alias", "expand", "permute", "transpose", "slice", "select", "squeeze", "unbind", "split", "t.default", "size", "stride", "empty", "vsde", "is_", "numel", "dim", "lift", "as_strided", "chunk")):
            ts = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor)]
            if ts and max(t.numel() for t in ts) >= 0:
                frames = [f for f in traceback.extract_stack() if "viforsdes_amd" in f.filename or "bench.py" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1])
                seen[(name, str([tuple(t.shape) for t in ts][:2]), where)] += 1
        return out
with Spy():
    tr._train_step(tr.ctx.model); tr.ctx.ema.update()
torch.cuda.synchronize()
for (name, shapes, where), n in sorted(seen.items(), key=lambda kv: -kv[1])[:150]:
    print(f"{n:3d} {name:32s} {shapes:48s} {where}")
