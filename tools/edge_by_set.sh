# k_edge_fused per edge set: durations of the decode-step launches of one rollout, classified by their position in the layer
# (temporal, map, agent repeat in that order); rocprofv3 kernel trace of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-1024}
rm -rf /tmp/kt
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/bench_with_lib.py --scenes $S --no-cpu-baseline --no-parity --no-strict --no-literal --steps 1 --warmup 1 > /tmp/kt.log 2>&1
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_edge_fused' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the decode-step launches: the most common grid size
grids = collections.Counter(r.get('Grid_Size_X', r.get('Grid_Size')) for r in rows)
g = grids.most_common(1)[0][0]
step = [r for r in rows if r.get('Grid_Size_X', r.get('Grid_Size')) == g]
# per rollout: 18 edgeless launches of the column-0 chain (same grid, short) then 16 x 18; drop launches shorter than 20 us
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in step]
long = [d for d in dur if d > 20.0]
print('launches', len(dur), 'of them in decode steps', len(long))
acc = collections.defaultdict(list)
for i, d in enumerate(long):
    acc[i % 3].append(d)
for k, name in enumerate(('temporal', 'map', 'agent')):
    v = acc[k]
    print(f'{name:9s} n {len(v):4d} avg us {sum(v)/len(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}')
PY
