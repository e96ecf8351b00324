# round 3, first GPU pass: GPU test suite, the default bench line, drop-in timing, kernel trace at 8 scenes (gap analysis)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
BENCH_VERBOSE=1 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r03a/bench.err | cut -c1-400
cut -c1-1500 gpurun_out/r03a/bench.json
python tools/bench_dropin.py 512 > gpurun_out/r03a/dropin.log 2>&1; tail -2 gpurun_out/r03a/dropin.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
INFGEN_GRAPH=0 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03a/kt8 -- python $R/bench.py --scenes 8 --no-cpu-baseline --no-parity --no-literal --steps 3 --warmup 2 > $R/gpurun_out/r03a/bench_kt8.log 2>&1
f=$(find $R/gpurun_out/r03a/kt8 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $f > $R/gpurun_out/r03a/gaps8.txt 2>&1
rm -rf $R/gpurun_out/r03a/kt8
head -60 $R/gpurun_out/r03a/gaps8.txt
