#!/bin/bash
# timing-only ablations of csrc/vsde_mlp.hip::deep256p_kernel (VSDE_DEEP256_ABL bits: 1 no MFMAs, 2 no activation loads, 4 no weight DMA,
# 8 no weight fragment reads, 16 no activation staging), one tools/deep_bench.py run each, tail launch off.   usage: tools/deep_ablate256.sh
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 8 16 6 24 30 31; do
  echo "== ABL $a"
  VSDE_DEEP256_TAIL=0 VSDE_DEEP256_ABL=$a python3 tools/deep_bench.py 2>/dev/null | grep "K=" | sed 's/HBM floor.*//'
done
