"""Which autograd nodes launch the small (non-vsde, non-hipBLASLt) kernels of the LV backward pass (GPU only)."""
import os, sys
from collections import defaultdict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from viforsdes_amd.examples.sdes import lv_problem

dev = torch.device("cuda:0")
tr = bench.build_trainer(lv_problem(), 512, dev, True, seed=1234)
model = tr.ctx.model
for _ in range(3):
    tr._train_step(model); tr.ctx.ema.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr._forward_backward(model)
    torch.cuda.synchronize()
evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
nodes = sorted([e for e in cpu if e.name.startswith("autograd::engine::evaluate_function")], key=lambda e: e.time_range.start)
agg = defaultdict(lambda: [0, 0.0])
fwd = defaultdict(lambda: [0, 0.0])
import bisect
starts = [n.time_range.start for n in nodes]
withk = [e for e in cpu if e.kernels]
print("cpu events", len(cpu), "with kernels", len(withk), "autograd nodes", len(nodes))
seen = set()
for e in sorted(withk, key=lambda e: e.time_range.end - e.time_range.start):   # innermost (shortest) op first
    ks = [k for k in e.kernels if "vsde" not in k.name and "Cijk" not in k.name and id(k) not in seen]
    for k in e.kernels:
        seen.add(id(k))
    if not ks:
        continue
    t = e.time_range.start
    i = bisect.bisect_right(starts, t) - 1
    inside = i >= 0 and nodes[i].time_range.start <= t <= nodes[i].time_range.end
    key = nodes[i].name.replace("autograd::engine::evaluate_function: ", "") if inside else "(forward) " + e.name
    tgt = agg if inside else fwd
    tgt[key][0] += len(ks); tgt[key][1] += sum(k.duration for k in ks)
print("backward: small kernels by autograd node")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{v[0]:4d} {v[1]:8.1f} us  {k}")
print("forward: small kernels by op")
for k, v in sorted(fwd.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{v[0]:4d} {v[1]:8.1f} us  {k}")
