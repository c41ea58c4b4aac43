#!/usr/bin/env python3
"""VariationalPosterior.sample(n) through the product API (EMA swap + captured call + copies of the results): calls per second
at the OU / LV benchmark sizes, fp32 (the reference's semantics) and mixed_precision=True.   python tools/probes/sample_api.py [lv|ou]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from viforsdes_amd.examples.sdes import lv_problem, ou_problem  # noqa: E402
from viforsdes_amd.posterior.variational_posterior import VariationalPosterior  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ou"
problem = lv_problem() if wl == "lv" else ou_problem()
n = 512 if wl == "lv" else 128
dev = torch.device("cuda:0")
tr = bench.build_trainer(problem, n, dev, True, seed=3)
vp = VariationalPosterior(model=tr.ctx.model, exponential_moving_average=tr.ctx.ema, prior=problem[3], observations=problem[1],
                          time_horizon=problem[4], time_step=problem[5], state_space=tr.state_space,
                          evidence_lower_bound_history=[], device=dev)
for mixed in (True, False):
    for _ in range(4):
        vp.sample(n, mixed_precision=mixed)
    torch.cuda.synchronize()
    reps = 30 if mixed else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        vp.sample(n, mixed_precision=mixed)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{wl} sample({n}, mixed_precision={mixed}): {1e3 * dt:.2f} ms per call = {n / dt:,.0f} paths/s (captured: {vp._captured.get((n, torch.bfloat16 if mixed else None)) is not None})")
