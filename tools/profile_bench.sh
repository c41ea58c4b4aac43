cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_prof.log 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof -name '*.db' | head -1) > $R/gpurun_out/r02_bench_lv_kernels_v6.txt 2>&1 || ls -R /tmp/prof | head
cd $R && python3 bench.py > gpurun_out/bench_r02_v6.json 2> gpurun_out/bench_r02_v6.err
tail -c 600 gpurun_out/bench_r02_v6.json
