cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-literal --no-strict --steps 3 --warmup 1 > $R/gpurun_out/prof/bench_kt.log 2>&1
find $R/gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof/kernel_stats.csv
find $R/gpurun_out/prof/kt -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/prof/pmc_$c -- python $R/bench.py --no-cpu-baseline --no-parity --no-literal --no-strict --steps 1 --warmup 1 > $R/gpurun_out/prof/bench_$c.log 2>&1
  f=$(find $R/gpurun_out/prof/pmc_$c -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f > $R/gpurun_out/prof/pmc_$c.summary.csv 2>&1
  python $R/tools/pmc_summary.py --by-grid $f > $R/gpurun_out/prof/pmc_$c.bygrid.csv 2>&1
  rm -rf $R/gpurun_out/prof/pmc_$c
done
tail -2 $R/gpurun_out/prof/bench_kt.log | cut -c1-300
head -8 $R/gpurun_out/prof/kernel_stats.csv | cut -c1-160
head -12 $R/gpurun_out/prof/pmc_FETCH_SIZE.summary.csv
