# SQ counters of k_edge_fused (24-bit rows) next to k_edge_mfma on the agent-set shaped probe (tools/probe_edge_mfma.py 1024)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_em
python $R/tools/probe_edge_mfma.py 1024 > $R/gpurun_out/prof_em/probe.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_em
  timeout -s KILL 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_em -- python $R/tools/probe_edge_mfma.py 1024 > $R/gpurun_out/prof_em/run_$tag.log 2>&1
  f=$(find /tmp/pmc_em -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_summary.py $f > $R/gpurun_out/prof_em/pmc_$tag.summary.csv 2>&1
  head -8 $R/gpurun_out/prof_em/pmc_$tag.summary.csv | cut -c1-220
done
rm -rf /tmp/ks_em
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_em -- python $R/tools/probe_edge_mfma.py 1024 > /dev/null 2>&1
f=$(find /tmp/ks_em -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/prof_em/kernel_stats.csv; head -6 $f | cut -c1-200
