# SQ counter pass of the default bench (separate run, no tracing): wave cycles, MFMA busy, VALU / wait breakdown per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_sq
  timeout -s KILL 500 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_sq -- python $R/bench.py --no-cpu-baseline --no-parity --no-literal --no-strict --steps 1 --warmup 1 > $R/gpurun_out/prof/bench_sq_$tag.log 2>&1
  f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f > $R/gpurun_out/prof/pmc_sq_$tag.summary.csv 2>&1
  head -5 $R/gpurun_out/prof/pmc_sq_$tag.summary.csv | cut -c1-200
done
