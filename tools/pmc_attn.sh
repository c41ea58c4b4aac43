# counters of the attention kernels at the LV encoder shape (tools/attnbench.py), one rocprofv3 pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  d=/tmp/pa_$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/attnbench.py > /dev/null 2>&1
  echo "== $set"
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) vsde::attn
done
