# PMC counters of the attention kernels at LV dims (tools/attn_core_bench.py) with the forward as one workgroup per pair
# (VSDE_ATTN_PERSIST=0) and as persistent workgroups (=1); one rocprofv3 pass per counter set, --pmc on its own.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for persist in 0 1; do
  export VSDE_ATTN_PERSIST=$persist
  echo "######## VSDE_ATTN_PERSIST=$persist"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    d=/tmp/paf_${persist}_$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/attn_core_bench.py > /dev/null 2>&1
    echo "== $set"
    python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) vsde::attn_
  done
done
