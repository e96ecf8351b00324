# fused edge kernel per edge set (temporal / map / agent): durations of the decode-step launches of one rollout from a rocprofv3
# kernel trace of a short bench run; usage: edge_by_set2.sh "<extra bench.py flags>" [kernel name substring]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${2:-k_edge_fused}
rm -rf /tmp/kt
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-strict --no-literal --steps 1 --warmup 1 $1 > /tmp/kt.log 2>&1
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) $K <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
grids = collections.Counter(r.get('Grid_Size_X', r.get('Grid_Size')) for r in rows)
g = grids.most_common(1)[0][0]
step = [r for r in rows if r.get('Grid_Size_X', r.get('Grid_Size')) == g]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in step]
long = [d for d in dur if d > 20.0]
print(sorted(set(r['Kernel_Name'][:60] for r in step)), 'launches', len(dur), 'in decode steps', len(long))
acc = collections.defaultdict(list)
for i, d in enumerate(long):
    acc[i % 3].append(d)
for k, name in enumerate(('temporal', 'map', 'agent')):
    v = acc[k]
    print(f'{name:9s} n {len(v):4d} avg us {sum(v)/len(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}')
PY
