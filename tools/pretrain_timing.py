#!/usr/bin/env python3
"""Seconds per pre-training run (reference trainer.py:208-259: B=4096 simulated paths per iteration) with the HIP
Euler-Maruyama simulator vs the torch time loop (both replayed from a HIP graph, as the trainer does).
    python tools/pretrain_timing.py [--iters 200] [--workload lv|ou]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_trainer  # noqa: E402
from viforsdes_amd import PretrainConfig  # noqa: E402
from viforsdes_amd.core import euler_maruyama as em  # noqa: E402
from viforsdes_amd.examples.sdes import lv_problem, ou_problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--workload", default="lv")
a = ap.parse_args()
problem = lv_problem() if a.workload == "lv" else ou_problem()
out = {}
for hip in (True, False):
    em.HIP_SIMULATOR = hip
    tr = build_trainer(problem, 16, torch.device("cuda:0"), True, seed=1, enc_hidden=64, enc_depth=1)
    torch.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best = tr.pretrain_sde_parameters(PretrainConfig(n_iterations=a.iters))
    torch.cuda.synchronize()
    out["hip" if hip else "torch_loop"] = (time.perf_counter() - t0, best.tolist())
for k, (s, best) in out.items():
    print(f"{a.workload} pretrain {a.iters} iterations x 4096 paths, {k}: {s:.2f} s ({1e3 * s / a.iters:.2f} ms/iteration), "
          f"best mean {[round(v, 3) for v in best]}")
