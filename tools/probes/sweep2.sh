R=$GRAFT_REPO_ROOT; cd $R
run() { for rep in 1 2; do env $1 python3 bench.py --no-cpu-baseline --no-ou --no-pmc --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.3f ms/step' % ('$1', d['ms_per_step']))"; done; }
run "VSDE_NOP=0"
for c in 1 2 3; do run "VSDE_COLSUM_CHUNKS=$c"; done
run "VSDE_COLSUM_CHUNKS=2 VSDE_ROWS_CHUNKS=3"
run "VSDE_COLSUM_CHUNKS=1 VSDE_ROWS_CHUNKS=3"
run "VSDE_NOP=1"
