# round 3 measurement pass: GPU tests, the default bench line, small-batch / insertion / C4 / C5 lines, rocprofv3 kernel stats,
# PMC traffic passes, SQ counter passes.  usage: bash tools/r03_profiles.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
BENCH_VERBOSE=1 python bench.py > $O/bench_s512.json 2> $O/bench_s512.err; echo "bench rc=$?"; cut -c1-400 $O/bench_s512.json
python bench.py --scenes 64 --no-cpu-baseline --no-literal --steps 10 > $O/bench_s64.json 2>/dev/null
python bench.py --scenes 8 --no-cpu-baseline --no-literal --steps 10 > $O/bench_s8.json 2>/dev/null
python bench.py --insertion --no-cpu-baseline --no-parity --steps 3 > $O/bench_ins_s512.json 2>/dev/null
python bench.py --insertion --rollout-steps 800 --scenes 128 --insert-headroom 320 --no-cpu-baseline --no-parity --steps 1 --warmup 1 > $O/bench_c4shape_s128.json 2>/dev/null
python bench.py --agents 256 --map-tokens 4096 --rollout-steps 800 --scenes 32 --no-cpu-baseline --no-parity --steps 1 --warmup 1 > $O/bench_c5shape_s32.json 2>/dev/null
for f in s64 s8 ins_s512 c4shape_s128 c5shape_s32; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms')"; done
python tools/bench_dropin.py 512 2>&1 | tail -1 | tee $O/dropin.log
bash tools/prof_round.sh > $O/prof_round.log 2>&1; tail -14 $O/prof_round.log | cut -c1-200
bash tools/prof_sq.sh > $O/prof_sq.log 2>&1; tail -16 $O/prof_sq.log | cut -c1-200
cp gpurun_out/prof/kernel_stats.csv $O/ 2>/dev/null; cp gpurun_out/prof/pmc_*.summary.csv $O/ 2>/dev/null
