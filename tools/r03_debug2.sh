cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
export PYTHONFAULTHANDLER=1
echo "== A: dropin 512"; timeout 600 python tools/bench_dropin.py 512 2>&1 | tail -25 | cut -c1-250
echo "== B: default bench"; BENCH_VERBOSE=1 timeout 900 python bench.py > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err; echo rc=$?; grep -v "^\[bench" gpurun_out/r03c/bench.err | tail -30 | cut -c1-250; grep "^\[bench" gpurun_out/r03c/bench.err | tail -4 | cut -c1-200
echo "== C: cu stream probe"; hipcc -O3 --offload-arch=gfx950 tools/cu_stream_probe.hip -o /tmp/cu_stream_probe 2>/dev/null && timeout 300 /tmp/cu_stream_probe 2>&1 | tee gpurun_out/r03c/cu_stream_probe.log
