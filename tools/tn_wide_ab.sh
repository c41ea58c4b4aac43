#!/bin/bash
# same-box A/B of the weight-gradient tiles of the head: fp32 MFMA form | split bf16 form (+ split weights)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${TW_AB_CONFIGS:-"VSDE_TW_SPLIT=0" "VSDE_TW_INTERLEAVE=0" "VSDE_TW_SPLIT=1" "VSDE_TW_WGS=512" "VSDE_TW_WGS=1536" "VSDE_TW_W1=100,VSDE_TW_W2=60" "VSDE_TW_W1=50,VSDE_TW_W2=30"}; do
  cfg=${cfg//,/ }
  d=/tmp/twab; rm -rf $d
  export $cfg
  timeout 150 rocprofv3 --kernel-trace -d $d -o p -- python3 $R/tools/head_ab.py > /tmp/twab.log 2>&1
  echo "== $cfg   $(tail -1 /tmp/twab.log)"
  python3 $R/tools/rocpd_stats.py $(find $d -name '*.db' | head -1) | grep -E "tn_wide|tn_grouped" 
  unset VSDE_TW_SPLIT VSDE_TW_WGS VSDE_TW_W1 VSDE_TW_W2 VSDE_TW_INTERLEAVE VSDE_TW_DBG
done
