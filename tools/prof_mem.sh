# memory-path counter passes of the default bench (separate runs, no tracing): L1 -> L2 read requests and their latency, L2 hit /
# miss, L2 -> fabric read requests with their occupancy (latency = LEVEL / RDREQ), DRAM credit stalls, texture-addresser stalls
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_REQ_sum TCC_STREAMING_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_mem
  timeout -s KILL 500 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_mem -- python $R/bench.py --no-cpu-baseline --no-parity --no-literal --no-strict --steps 1 --warmup 1 > $R/gpurun_out/prof/bench_mem_$tag.log 2>&1
  f=$(find /tmp/pmc_mem -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f > $R/gpurun_out/prof/pmc_mem_$tag.summary.csv 2>&1
  head -6 $R/gpurun_out/prof/pmc_mem_$tag.summary.csv | cut -c1-260
done
