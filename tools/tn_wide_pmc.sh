#!/bin/bash
# L2 behaviour of the head weight-gradient kernel (one rocprofv3 pass per counter set, no other trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0 64; do
  export VSDE_TW_DBG=$dbg
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
    d=/tmp/pmc_tw; rm -rf $d
    timeout 120 rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/head_probe.py 3 > /dev/null 2>&1
    echo "== dbg=$dbg  $set"
    python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) tn_wide_split
  done
done
