#!/usr/bin/env python3
"""Call sites of the aten operators of one LV training step (forward, loss, optimizer glue -- the thread that calls the step; the
autograd engine's device thread is not seen): a TorchDispatchMode that records, for every operator that launches work on the GPU, the
innermost frame inside the package.    python tools/step_op_sites.py [lv|ou]"""
import os, sys, traceback, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
from viforsdes_amd.examples.sdes import lv_problem, ou_problem

which = sys.argv[1] if len(sys.argv) > 1 else "lv"
dev = torch.device("cuda:0")
tr = bench.build_trainer(lv_problem() if which == "lv" else ou_problem(), 512 if which == "lv" else 128, dev, True, seed=1234)
model = tr.ctx.model
for _ in range(3):
    tr._train_step(model)
torch.cuda.synchronize()
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.slice",
        "aten.select", "aten.as_strided", "aten._unsafe_view", "aten.permute", "aten.split", "aten.chunk", "aten.unbind", "aten.reshape",
        "aten.is_", "aten.sym_", "aten.size", "aten.stride", "aten.empty", "aten.lift_fresh", "aten._local_scalar_dense", "aten.item",
        "aten.unflatten", "aten.narrow", "aten.view_as", "prim.", "aten.result_type", "aten.set_")
sites = collections.Counter()

class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(name.startswith(s) for s in SKIP):
            frames = [f for f in traceback.extract_stack() if "viforsdes_amd" in f.filename and "tools/" not in f.filename]
            fr = frames[-1] if frames else None
            outer = next((f for f in reversed(frames) if "primitives/fused.py" not in f.filename and "_hip.py" not in f.filename), fr)
            key = (name, f"{outer.filename.split('viforsdes_amd/')[-1]}:{outer.lineno} {outer.name}" if outer else "?")
            sites[key] += 1
        return func(*args, **(kwargs or {}))

with Log():
    tr._train_step(model)
torch.cuda.synchronize()
print(f"{which}: {sum(sites.values())} operator calls on the stepping thread")
by_site = collections.defaultdict(list)
for (name, site), c in sites.items():
    by_site[site].append((c, name))
for site, ops in sorted(by_site.items(), key=lambda kv: -sum(c for c, _ in kv[1])):
    print(f"{sum(c for c, _ in ops):4d}  {site}")
    for c, name in sorted(ops, reverse=True):
        print(f"        {c:3d} {name}")
