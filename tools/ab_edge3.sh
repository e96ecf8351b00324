#!/bin/bash
# A/B of the fused edge kernel's lane layouts on the headline batch: k_edge_fused (0) vs k_edge_fused3 (1) with 4 / 6 / 8 edges per trip,
# and k_edge_fused3 with the next trip's rhat rows prefetched (INFGEN_EDGE3_PF=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab3
INFGEN_EDGE3_PF=1 python -m pytest tests/test_ops_gpu.py -q -k "edge_fused3" 2>&1 | tail -3
for cfg in "0 6 0" "1 6 0" "1 4 1"; do
  set -- $cfg
  INFGEN_EDGE3_PF=$3 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-literal --no-strict --steps 5 --warmup 2 --edge-kernel $1 --edge-loop $2 > gpurun_out/ab3/k$1_g$2_p$3.json 2> gpurun_out/ab3/k$1_g$2_p$3.err
  python - <<PY
import json
d = json.load(open('gpurun_out/ab3/k$1_g$2_p$3.json'))
r = d['roofline']
print('edge_kernel $1 G $2 PF $3: value %.3f M  ms %.2f  frac %.4f  edge avg_launch_us %.1f  per-kernel' % (d['value'] / 1e6, d['ms_per_step'], r['frac'], r['avg_launch_us']), {k: v for k, v in r['per_kernel_ms_one_rollout'].items() if v > 2})
PY
done
