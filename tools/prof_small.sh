# rocprofv3 kernel stats of the small-batch lines (8 and 64 scenes per GPU): bash tools/prof_small.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03b}
mkdir -p $R/gpurun_out/$TAG
for sc in 8 64; do
  rm -rf /tmp/kt$sc
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$sc -- python $R/bench.py --scenes $sc --no-cpu-baseline --no-parity --no-literal --steps 5 --warmup 2 > $R/gpurun_out/$TAG/bench_kt_s$sc.log 2>&1
  find /tmp/kt$sc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/$TAG/kernel_stats_s$sc.csv
  head -9 $R/gpurun_out/$TAG/kernel_stats_s$sc.csv | cut -c1-150
done
