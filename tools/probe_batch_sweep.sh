# round-4 probes: scenes sweep of the headline workload, host profile of the drop-in entry, kernel trace of an insertion rollout
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
for s in 128 171 256 384 768 1024; do
  timeout 300 python bench.py --scenes $s --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-literal --no-strict > gpurun_out/sweep/sweep_s$s.json 2> gpurun_out/sweep/sweep_s$s.err
  python -c "
import json
d=json.load(open('gpurun_out/sweep/sweep_s$s.json')); print('scenes', $s, round(d['value']/1e6,3),'M', round(d['ms_per_step'],2),'ms', 'edge avg us', round(d['roofline'].get('avg_launch_us',0),1), 'frac', round(d['roofline']['frac'],4))"
done
timeout 400 python tools/host_profile_dropin.py 512 > gpurun_out/sweep/dropin_profile.txt 2>&1
head -60 gpurun_out/sweep/dropin_profile.txt
timeout 600 bash tools/ins_trace.sh 512 > gpurun_out/sweep/ins_trace.txt 2>&1
head -50 gpurun_out/sweep/ins_trace.txt
