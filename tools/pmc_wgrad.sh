# per-shape durations of the weight-gradient kernels (kernel trace), then the counters that explain them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python3 $R/tools/wgrad_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
for r in rows:
    if 'wgrad' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:40],) + tuple(r[c] for c in r if c.startswith('Grid_Size'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, 'n=%d median=%.1f us min=%.1f' % (len(v), v[len(v) // 2], v[0]))
PY
