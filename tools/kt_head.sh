# per-kernel durations of the head alone (tools/head_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/kth -o k -- python3 $R/tools/head_probe.py 5 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py $(find /tmp/kth -name '*.db' | head -1) | head -${1:-14} | cut -c1-60,90-150
