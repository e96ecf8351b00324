#!/bin/bash
# same-box A/B of environment switches on the default bench: tools/ab_env.sh "VAR=a" "VAR=b" ...   ("-" = no variable)
for v in "$@"; do
  if [ "$v" = "-" ]; then pre=""; else pre="$v"; fi
  env $pre timeout 400 python bench.py --no-cpu-baseline --no-literal --no-strict --no-parity --steps 5 2>/dev/null > /tmp/ab_env.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_env.json'))
r = d['roofline']
print(sys.argv[1], 'value', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'edge launch us', round(r['avg_launch_us'], 1), 'frac', round(r['frac'], 4))
PY
done
