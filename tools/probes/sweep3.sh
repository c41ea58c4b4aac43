R=$GRAFT_REPO_ROOT; cd $R
run() { for rep in 1 2; do env $1 python3 bench.py --no-cpu-baseline --no-ou --no-pmc --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.3f ms/step' % ('$1', d['ms_per_step']))"; done; }
run "VSDE_NOP=0"
for n in 48 64 96 128; do run "VSDE_WGRAD_NSPLIT=$n"; done
run "VSDE_NOP=1"
