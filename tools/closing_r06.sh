# round 6 closing numbers on ONE box: kernel trace + FETCH / WRITE passes (prof_round), SQ passes (prof_sq), the default line,
# the per-set launch durations of the fused edge kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; mkdir -p $O
bash tools/prof_round.sh > $O/prof_round.log 2>&1
bash tools/prof_sq.sh > $O/prof_sq.log 2>&1
bash tools/edge_by_set2.sh "" k_edge_fused3 > $O/edge_by_set.txt 2>&1
timeout 900 python bench.py > $O/bench_s1024.json 2> $O/bench_s1024.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/suite/bench_s1024.json'))
r = d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_us', r['avg_launch_us'], 'traffic_ratio', r['traffic_ratio'])
print({k: d[k] for k in d if k.endswith('_value') or k.endswith('_ms') or k.startswith('multi')})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity']['ok'])
PY
head -4 gpurun_out/prof/kernel_stats.csv | cut -c1-150
cat $O/edge_by_set.txt
