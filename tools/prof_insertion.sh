#!/bin/bash
# GPU box: rocprofv3 kernel stats of the insertion-on rollout at 1024 scenes (where does the second half of its time go?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_ins
timeout -s KILL 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ins/kt -- python $R/bench.py --insertion --no-cpu-baseline --no-parity --no-literal --no-strict --steps 2 --warmup 1 > $R/gpurun_out/prof_ins/bench_kt.log 2>&1
find $R/gpurun_out/prof_ins/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_ins/kernel_stats.csv
f=$(find $R/gpurun_out/prof_ins/kt -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/prof_ins/by_grid.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Workgroup_Size_X', ''))
    a = agg[k]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{k[0]:60s} grid {k[1]:>9s} wg {k[2]:>5s} calls {a[0]:6d} total {a[1] / 1e3:9.2f} ms avg {a[1] / a[0]:9.1f} us {100 * a[1] / tot:5.1f} %')
PY
rm -rf $R/gpurun_out/prof_ins/kt
tail -1 $R/gpurun_out/prof_ins/bench_kt.log | cut -c1-200
head -25 $R/gpurun_out/prof_ins/by_grid.txt
