cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_ou -o b -- python3 $R/bench.py --workload ou --batch 128 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_ou -name '*.db' | head -1) | head -40 | cut -c1-70,88-150
