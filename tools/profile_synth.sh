cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_sy -o b -- python3 $R/bench.py --workload synthetic --batch 256 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_sy -name '*.db' | head -1) > $R/gpurun_out/r02_bench_synth_kernels_v2.txt 2>/dev/null
head -32 $R/gpurun_out/r02_bench_synth_kernels_v2.txt | cut -c1-75,88-150
