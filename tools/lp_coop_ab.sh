#!/bin/bash
# k_layers_p launched cooperatively (INFGEN_LAYERS_P=2, opt-in) vs plainly (INFGEN_LAYERS_P=1, the default): ms per rollout at 8 / 64 scenes
for c in 2 1; do for s in 8 64; do
  INFGEN_LAYERS_P=$c timeout 300 python bench.py --scenes $s --no-cpu-baseline --no-literal --no-strict --no-parity --steps 30 2>/dev/null > /tmp/lp_ab.json
  python - "$c" "$s" <<'PY'
import json, sys
d = json.load(open('/tmp/lp_ab.json'))
print('coop', sys.argv[1], 'scenes', sys.argv[2], round(d['value']), round(d['ms_per_step'], 3))
PY
done; done
