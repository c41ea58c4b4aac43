#!/bin/bash
# (A/B switches exist only in the tools' library: python -m viforsdes_amd.build --ablations)
export VSDE_HIP_LIB=${VSDE_HIP_LIB:-$GRAFT_REPO_ROOT/viforsdes_amd/libvsde_hip_abl.so}
# per-kernel durations of the attention core (tools/attn_core_bench.py under rocprofv3) with the wide dq kernel off / on, one box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 0 1; do
  export VSDE_ATTN_DQ_WIDE=$s; echo "== VSDE_ATTN_DQ_WIDE=$s"
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o b -- python3 $R/tools/attn_core_bench.py > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/prof_ab -name '*.db' | head -1) | grep -E "attn_bwd" | cut -c1-70,90-150
done
