cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ea = [r for r in rows if 'k_edge_attn' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ea]
print('edge_attn launches', len(d))
for k, name in enumerate(('temporal', 'map', 'agent')):
    sel = d[k::3]
    sel_nz = [x for x in sel if x > 30]
    print(name, 'n', len(sel), 'mean us', sum(sel) / len(sel), 'mean of >30us', sum(sel_nz) / max(1, len(sel_nz)), 'max', max(sel))
ah = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'k_attn_h' in r['Kernel_Name']]
print('attn_h n', len(ah), 'mean', sum(ah) / len(ah))
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows[:-1], rows[1:])]
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('total span ms', tot / 1e6, 'sum of positive gaps ms', sum(g for g in gaps if g > 0) / 1e6, 'n kernels', len(rows))
PY
