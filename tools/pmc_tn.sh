# counters of the grouped weight-gradient reduction of the head (tools/head_probe.py), one rocprofv3 pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-tn_grouped}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pt_$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/head_probe.py 3 > /dev/null 2>&1
  echo "== $set"
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) $K
done
