# kernel trace of one rollout with insertion on, grouped by (kernel, grid size): where the sub-loop's time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-512}
rm -rf /tmp/kt
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --insertion --scenes $S --no-cpu-baseline --no-parity --no-strict --no-literal --steps 1 --warmup 1 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-200
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ig::', ''), r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    acc[k][0] += 1; acc[k][1] += d
busy = sum(v[1] for v in acc.values())
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels', len(rows), 'span ms', tot / 1e6, 'busy ms', busy / 1e6, '(4 rollouts: warm-up, profile leg, timed, + retries)')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{k[0][:52]:52s} grid {k[1]:>8s} wg {k[2]:>5s} n {v[0]:6d} total ms {v[1]/1e6:9.2f} avg us {v[1]/v[0]/1e3:8.1f}')
PY
