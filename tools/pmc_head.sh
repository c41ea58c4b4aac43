# HBM-side counters (FETCH_SIZE / WRITE_SIZE, separate passes) and issue counters of every kernel of the head path at LV size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  d=/tmp/ph_$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $d -o p -- python3 $R/tools/head_probe.py 3 > /dev/null 2>&1
  echo "== $set   (FETCH_SIZE / WRITE_SIZE in KB; gfx950: double FETCH_SIZE for wide coalesced reads, MI355X_MICROARCH.md)"
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) vsde::
done
