#!/bin/bash
# (A/B switches exist only in the tools' library: python -m viforsdes_amd.build --ablations)
export VSDE_HIP_LIB=${VSDE_HIP_LIB:-$GRAFT_REPO_ROOT/viforsdes_amd/libvsde_hip_abl.so}
# In-step sweep of the launch-shape switches with the driver's bench command: tuning decisions taken in one round go
# stale when the kernels change in the next.   usage (GPU box): tools/knob_sweep.sh [lv|ou] > gpurun_out/rNN/knobs.txt
R=$GRAFT_REPO_ROOT; cd $R
WL=${1:-lv}
run() {  # run "<env assignments>"
  for rep in 1 2; do
    env $1 python3 bench.py --workload $WL --no-cpu-baseline --no-ou --no-pmc --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.3f ms/step' % ('$1', d['ms_per_step']))"
  done
}
run "VSDE_NOP=0"
for c in 1 2 3 4; do run "VSDE_ROWS_CHUNKS=$c"; done
for t in 128 256; do run "VSDE_WGRAD_TN=$t"; done
for n in 8 16 24 32; do run "VSDE_WGRAD_NSPLIT=$n"; done
for c in 2 4 8; do run "VSDE_COLSUM_CHUNKS=$c"; done
run "VSDE_ROWS_TAIL=0"
run "VSDE_ROWS_XSTAGE=0"
run "VSDE_HEAD_MP=8"
run "VSDE_NOP=1"
