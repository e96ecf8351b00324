cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
python -m pytest tests/test_rollout_gpu.py tests/test_ops_gpu.py tests/test_torch_ops_gpu.py tests/test_modules_gpu.py -m gpu -x -q 2>&1 | tail -4
for sc in 8 64; do for nf in 1 0; do INFGEN_NO_TAIL_FOLD=$nf python bench.py --scenes $sc --no-cpu-baseline --no-parity --no-literal --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('scenes $sc no_fold=$nf', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms')"; done; done
python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r03n/bench_s512.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03n/bench_s512.json')); print(round(d['value']/1e6,3),'M', round(d['ms_per_step'],2),'ms', d['config']['c3_literal'], d['parity']['ok'])"
python tools/bench_dropin.py 512 2>&1 | tail -1
