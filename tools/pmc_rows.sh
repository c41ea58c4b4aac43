cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  d=/tmp/pr_$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/linear_bench.py > /dev/null 2>&1
  echo "== $set"
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) lin_rows_kernel
done
