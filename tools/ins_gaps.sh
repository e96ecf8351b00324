# idle time between consecutive kernels of an insertion rollout, by the kernel in front of the gap (tools/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-512}
rm -rf /tmp/kt
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --insertion --scenes $S --no-cpu-baseline --no-parity --no-strict --no-literal --steps 2 --warmup 1 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-200
python $R/tools/trace_gaps.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) | head -60
