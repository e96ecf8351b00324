cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
for s in 8 64; do timeout 300 python bench.py --scenes $s --no-cpu-baseline --no-literal --no-strict > gpurun_out/r04c/bench_s$s.json 2> gpurun_out/r04c/bench_s$s.err; done
timeout 500 python bench.py --insertion --no-cpu-baseline --no-literal --no-strict > gpurun_out/r04c/bench_ins_s512.json 2> gpurun_out/r04c/bench_ins_s512.err
timeout 900 python bench.py --insertion --rollout-steps 800 --scenes 128 --insert-headroom 320 --steps 2 --warmup 1 --no-cpu-baseline --no-literal --no-strict > gpurun_out/r04c/bench_c4shape_s128.json 2> gpurun_out/r04c/bench_c4shape_s128.err
timeout 900 python bench.py --agents 256 --map-tokens 4096 --rollout-steps 800 --scenes 32 --steps 2 --warmup 1 --no-cpu-baseline --no-literal --no-strict > gpurun_out/r04c/bench_c5shape_s32.json 2> gpurun_out/r04c/bench_c5shape_s32.err
timeout 300 python tools/bench_dropin.py 512 > gpurun_out/r04c/dropin.log 2>&1
for f in gpurun_out/r04c/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value']/1e6,3),'M', round(d['ms_per_step'],2),'ms', d['config'].get('agents_inserted_last_rollout'))"; done
tail -1 gpurun_out/r04c/dropin.log
