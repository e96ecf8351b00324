# per-edge-set launch durations of the edge kernel (temporal / map / agent) from a kernel trace, for a list of
# "<edge-loop> <INFGEN_EDGE_DBG>" configurations; usage (on the GPU box): bash tools/edge_probe.sh "2 0" "2 1" "2 2"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  rm -rf /tmp/kt
  INFGEN_EDGE_DBG=$2 timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --edge-loop $1 $EXTRA > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
  echo "== edge-loop $1 dbg $2"
  python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
ea = [dur(r) for r in rows if 'k_edge_fused' in r['Kernel_Name'] or 'k_edge_attn(' in r['Kernel_Name']]
# the decode layers launch temporal, map, agent in turn; the 18 launches of the edgeless column-0 chain come first
ea = [x for x in ea]
for k, name in enumerate(('temporal', 'map', 'agent')):
    sel = [x for x in ea[k::3]]
    big = sorted(sel)[len(sel) // 4:]
    print(f'{name:9s} n {len(sel):4d} mean {sum(sel) / len(sel):7.1f} us  upper-3/4 mean {sum(big) / len(big):7.1f}  max {max(sel):7.1f}')
for kn in ('k_attn_h', 'k_fourier_h', 'k_edge_fused', 'k_edge_attn('):
    d = [dur(r) for r in rows if kn in r['Kernel_Name']]
    if d:
        print(f'{kn:14s} n {len(d):4d} total {sum(d) / 1e3:7.2f} ms mean {sum(d) / len(d):7.1f} us')
PY
done
