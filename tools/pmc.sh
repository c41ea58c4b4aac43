#!/bin/bash
# PMC counters of the kernels one driver script launches, ONE rocprofv3 pass per counter set (never combined with --stats or the
# hip/hsa trace domains), per-kernel means printed by tools/pmc_summary.py.
#   usage (on the GPU box): tools/pmc.sh <kernel-name filter> <driver.py> [driver args...]
#   e.g.  tools/pmc.sh vsde::head tools/head_probe.py 3            (serial GRU kernels at LV size -> profiles/rNN_pmc_head_lv.txt)
#         tools/pmc.sh vsde::attn tools/attn_core_bench.py         (attention core at LV dims)
#         tools/pmc.sh lin_rows tools/linear_probe.py 1536 256 ;  tools/pmc.sh wgrad tools/wgrad_bench.py ; tools/pmc.sh tn_wide tools/head_probe.py 3
# FETCH_SIZE / WRITE_SIZE are in KiB summed over the XCDs; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE undercounts wide coalesced
# reads by 2x (the serial GRU kernels read with 4-byte lanes: raw counter).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=$1; shift
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  d=/tmp/pmc_$(echo $set | tr ' ' '_' | cut -c1-30)
  rm -rf $d
  rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/$@ > /dev/null 2>&1
  echo "== $set"
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) $F
done
