#!/bin/bash
# (A/B switches exist only in the tools' library: python -m viforsdes_amd.build --ablations)
export VSDE_HIP_LIB=${VSDE_HIP_LIB:-$GRAFT_REPO_ROOT/viforsdes_amd/libvsde_hip_abl.so}
# per-kernel durations of the attention core (tools/attn_core_bench.py under rocprofv3).  A/B on ONE box:
#   tools/attn_bwd_kernels.sh                      VSDE_ATTN_SPLIT = 0 and 1 with the tree's library
#   tools/attn_bwd_kernels.sh base                 ... and first with viforsdes_amd/libvsde_hip_base.so (a library built from another commit) swapped in
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o b -- python3 $R/tools/attn_core_bench.py > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/prof_ab -name '*.db' | head -1) | grep -E "attn_bwd|attn_fwd" | cut -c1-60,90-150
}
if [ "$1" = base ]; then
  cp $R/viforsdes_amd/libvsde_hip.so /tmp/libvsde_new.so
  cp $R/viforsdes_amd/libvsde_hip_base.so $R/viforsdes_amd/libvsde_hip.so
  echo "== base library"; run
  cp /tmp/libvsde_new.so $R/viforsdes_amd/libvsde_hip.so
fi
for s in 0 1; do export VSDE_ATTN_SPLIT=$s; echo "== VSDE_ATTN_SPLIT=$s"; run; done
