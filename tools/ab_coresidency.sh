#!/bin/bash
# two engines on two HIP streams with the edge kernel capped at ONE 75 KB workgroup per CU (dynamic LDS pad),
# so that the other stream's node kernel (k_attn_h: ~50 KB per workgroup) can be co-resident with it
cd $GRAFT_REPO_ROOT
for pad in 0 10240; do
  INFGEN_EDGE_LDS_PAD=$pad timeout 400 python bench.py --no-cpu-baseline --no-parity --no-literal --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pad $pad: one stream %.2f M (%.1f ms), two streams %.2f M (%.1f ms)' % (d['value']/1e6, d['ms_per_step'], d['two_streams_value']/1e6, d['config']['two_streams']['ms_per_step']))"
done
