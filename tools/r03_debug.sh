cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
export PYTHONFAULTHANDLER=1
echo "== A: bench 64 scenes, no literal, graph off"; INFGEN_GRAPH=0 timeout 300 python bench.py --scenes 64 --no-cpu-baseline --no-parity --no-literal --steps 2 2>&1 | tail -3 | cut -c1-300
echo "== B: bench 64 scenes, no literal, graph on (auto)"; timeout 300 python bench.py --scenes 64 --no-cpu-baseline --no-parity --no-literal --steps 2 2>&1 | tail -12 | cut -c1-300
echo "== C: bench 512, literal legs, serialized"; AMD_SERIALIZE_KERNEL=3 timeout 600 python bench.py --no-cpu-baseline --no-parity --steps 2 2>&1 | tail -12 | cut -c1-300
echo "== D: hazard repro"; hipcc -O3 --offload-arch=gfx950 tools/hazard_repro.hip -o /tmp/hazard_repro 2>/dev/null && timeout 300 /tmp/hazard_repro 300 600 2>&1 | tee gpurun_out/r03b/hazard_repro.log | tail -8
echo "== E: gpu tests per file"
for f in tests/test_*gpu.py; do timeout 1200 python -m pytest $f -q -m gpu -x > gpurun_out/r03b/$(basename $f).log 2>&1; echo "$f rc=$? $(tail -1 gpurun_out/r03b/$(basename $f).log | cut -c1-150)"; done
