cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-512}
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --insertion --scenes $S --no-cpu-baseline --steps 1 --warmup 1 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-140
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels', len(rows), 'span ms', tot / 1e6, 'busy ms', busy / 1e6)
PY
