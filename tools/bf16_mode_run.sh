#!/bin/bash
# GPU box: the bf16-operand mode (gemm_terms = 2) - its C5-shape test and the C5-shape bench line under the three arithmetics
# (64 scenes x 256 agents = 16 k rows per launch: every node launch takes the split kernels by size)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_baseline_shapes_gpu.py -q -s -m gpu -k "c5_shape_bf16 or c5_shape_reduced" 2>&1 | grep -v "^$" | tail -15
for t in 3 2 1; do
  timeout 600 python bench.py --agents 256 --map-tokens 4096 --rollout-steps 800 --scenes 64 --steps 2 --warmup 1 --gemm-terms $t \
    --no-parity --no-strict --no-cpu-baseline > gpurun_out/bench_c5_s64_terms$t.json 2> gpurun_out/bench_c5_s64_terms$t.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_c5_s64_terms$t.json').read().strip().splitlines()[-1])
print('gemm_terms', $t, d['value'], d['unit'], d['ms_per_step'], 'ms', d['dtype'])
PY
done
