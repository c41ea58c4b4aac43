#!/bin/bash
# same-box A/B of two builds of the library with the driver's bench command:
#   tools/ab_bench.sh [bench args]     runs bench.py with viforsdes_amd/libvsde_hip_base.so swapped in, then with the tree's library, twice each
R=$GRAFT_REPO_ROOT
cd $R
cp viforsdes_amd/libvsde_hip.so /tmp/libvsde_new.so
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then cp viforsdes_amd/libvsde_hip_base.so viforsdes_amd/libvsde_hip.so; else cp /tmp/libvsde_new.so viforsdes_amd/libvsde_hip.so; fi
    python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['ms_per_step'],3), 'ms/step; head fwd', round(d['roofline']['avg_ms']*1e3), 'us frac', round(d['roofline']['frac'],3), '; ou', (d.get('ou') or {}).get('ms_per_step'))"
  done
done
cp /tmp/libvsde_new.so viforsdes_amd/libvsde_hip.so
