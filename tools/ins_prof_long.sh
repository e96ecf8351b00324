# kernel stats of a long rollout with insertion and ample head-room: bash tools/ins_prof_long.sh [scenes] [R] [headroom]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-128}; RS=${2:-400}; H=${3:-320}
rm -rf /tmp/ktl
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktl -- python $R/bench.py --insertion --scenes $S --rollout-steps $RS --insert-headroom $H --no-cpu-baseline --steps 1 --warmup 0 > /tmp/ktl.log 2>&1
f=$(find /tmp/ktl -name "*kernel_stats.csv" | head -1)
head -16 $f | cut -c1-150
python - $(find /tmp/ktl -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
tot = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels', len(rows), 'span ms', tot / 1e6, 'busy ms', busy / 1e6)
PY
