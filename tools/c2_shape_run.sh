#!/bin/bash
# GPU box: the bench line at BASELINE C2's shape (32 agents, 512 map tokens, R = 80) for 1024 and 2048 scenes per GPU
mkdir -p gpurun_out
for s in 1024 2048; do
  timeout 600 python bench.py --agents 32 --map-tokens 512 --scenes $s --steps 5 --warmup 2 --no-cpu-baseline --no-literal > gpurun_out/bench_c2shape_s$s.json 2> gpurun_out/bench_c2shape_s$s.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_c2shape_s$s.json').read().strip().splitlines()[-1])
r = d['roofline']
print($s, d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r.get('traffic'), d.get('parity', {}).get('ok'))
PY
done
