#!/bin/bash
# full ELBO step per path at batch sizes around whole rounds of 128-row workgroups (rows = batch * 401; 512 resident workgroups)
cd $GRAFT_REPO_ROOT
for b in 448 480 490 500 512 530 560 600 654; do
  timeout 300 python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-ou --no-pmc --no-families 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=$b
print(f'batch {b:4d}  rows {b*401:7d}  blocks {b*401/128:7.1f}  rounds {b*401/128/512:5.2f}   {d[\"ms_per_step\"]:7.3f} ms/step   {d[\"ms_per_step\"]/b*1e3:6.2f} us per path')"
done
