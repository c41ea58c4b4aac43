#!/bin/bash
# per-kernel durations of the attention forward (tools/attn_core_bench.py under rocprofv3), the tree's library against
# viforsdes_amd/libvsde_hip_base.so (a library built from another commit), alternating twice on one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in "" "$R/viforsdes_amd/libvsde_hip_base.so"; do
    export VSDE_HIP_LIB=$lib; echo "== ${lib:-tree}"
    rm -rf /tmp/prof_ab
    rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o b -- python3 $R/tools/attn_core_bench.py > /dev/null 2>&1
    python3 $R/tools/rocpd_stats.py $(find /tmp/prof_ab -name '*.db' | head -1) | grep -E "attn_fwd" | cut -c1-70,90-150
  done
done
