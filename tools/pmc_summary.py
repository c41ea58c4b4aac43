#!/usr/bin/env python3
"""Per-kernel means of the PMC counters in a rocprofv3 rocpd database (``--pmc`` run, sqlite).
    python tools/pmc_summary.py gpurun_out/x/p_results.db [kernel-name-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
dur = defaultdict(list)
for name, disp, ctr, val, dt in db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    if flt not in name:
        continue
    key = name[:80]
    acc[key][ctr] += val          # sum over the hardware instances (XCDs / SEs) of a dispatch
    cnt[key].add(disp)
    dur[key].append(dt)
for key in acc:
    n = len(cnt[key])
    print(f"{key}  dispatches={n} avg_duration_us={sum(dur[key]) / len(dur[key]) / 1e3:.1f}")
    for ctr, v in sorted(acc[key].items()):
        print(f"    {ctr:32s} mean={v / n:.4g}")
