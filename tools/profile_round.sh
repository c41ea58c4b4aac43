#!/bin/bash
# rocprofv3 kernel trace (per-kernel durations) of the bench workloads; summaries under gpurun_out/<round>/, to be copied into profiles/.
#   usage (on the GPU box): tools/profile_round.sh <round> <tag> [lv] [ou] [synthetic]      e.g.  tools/profile_round.sh r04 v1 lv ou
# Per workload: the kernel table of a profiled `bench.py --workload W --no-cpu-baseline --no-ou` run, then the same command unprofiled
# (its JSON line).  rocprofv3 needs the program itself after `--` (python3, no shell / env hop) and TMPDIR on /tmp.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${1:-r04}; TAG=${2:-v1}; shift 2
WL=${@:-lv}
O=$R/gpurun_out/$ROUND
mkdir -p $O
for w in $WL; do
  case $w in
    lv) args="--workload lv" ;;
    ou) args="--workload ou" ;;
    synthetic) args="--workload synthetic --steps 5 --warmup 2" ;;
  esac
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o b -- python3 $R/bench.py $args --no-cpu-baseline --no-ou --no-pmc > $O/bench_prof_${w}_$TAG.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/prof_$w -name '*.db' | head -1) > $O/${ROUND}_bench_${w}_kernels_$TAG.txt 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/prof_$w -name '*.db' | head -1) busy >> $O/${ROUND}_bench_${w}_kernels_$TAG.txt 2>&1
  (cd $R && python3 bench.py $args --no-cpu-baseline --no-ou --detail-out $O/bench_${ROUND}_${w}_$TAG.json > $O/bench_${ROUND}_${w}_${TAG}_headline.json 2> $O/bench_${ROUND}_${w}_$TAG.err)
  python3 -c "
import json
d=json.loads(open('$O/bench_${ROUND}_${w}_$TAG.json').read()); print('$w', round(d['ms_per_step'],3),'ms/step', round(d['value']), 'paths/s', round(d['sampled_paths_per_sec']), 'sampled/s, head only', round(d['sampled_paths_per_sec_head_only']))
"
done
