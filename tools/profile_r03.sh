# rocprofv3 kernel trace of the default LV bench run + the synthetic (config 5) and OU steps; summaries under gpurun_out/r03/
# usage: tools/profile_r03.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-v1}
mkdir -p $R/gpurun_out/r03
rocprofv3 --kernel-trace --stats -d /tmp/prof_lv -o b -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r03/bench_prof_lv_$T.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_lv -name '*.db' | head -1) > $R/gpurun_out/r03/r03_bench_lv_kernels_$T.txt 2>&1 || ls -R /tmp/prof_lv | head
rocprofv3 --kernel-trace --stats -d /tmp/prof_sy -o b -- python3 $R/bench.py --workload synthetic --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r03/bench_prof_synth_$T.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_sy -name '*.db' | head -1) > $R/gpurun_out/r03/r03_bench_synth_kernels_$T.txt 2>&1
cd $R
python3 bench.py --workload synthetic --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03/bench_r03_synth_$T.json 2> gpurun_out/r03/bench_r03_synth_$T.err
python3 bench.py --workload ou --no-cpu-baseline > gpurun_out/r03/bench_r03_ou_$T.json 2> gpurun_out/r03/bench_r03_ou_$T.err
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench_r03_lv_$T.json 2> gpurun_out/r03/bench_r03_lv_$T.err
python3 -c "
import json
for w in ('lv','synth','ou'):
    d=json.loads(open('gpurun_out/r03/bench_r03_%s_$T.json'%w).read()); print(w, round(d['ms_per_step'],2),'ms/step', round(d['value']), 'paths/s', round(d['sampled_paths_per_sec']), 'sampled/s')
"
