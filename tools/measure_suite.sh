# the round's closing measurements on the final build: profiles (kernel stats + PMC passes of the default workload), the default
# bench line, the secondary lines, the drop-in entry, a determinism soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; mkdir -p $O
bash tools/prof_round.sh > $O/prof_round.log 2>&1
bash tools/prof_sq.sh > $O/prof_sq.log 2>&1
B="--no-cpu-baseline --no-literal --no-strict"
timeout 900 python bench.py > $O/bench_s1024.json 2> $O/bench_s1024.err
timeout 600 python bench.py --scenes 512 $B > $O/bench_s512.json 2> $O/bench_s512.err
for s in 8 64; do timeout 300 python bench.py --scenes $s --steps 20 --warmup 3 $B > $O/bench_s$s.json 2> $O/bench_s$s.err; done
timeout 600 python bench.py --insertion --scenes 512 $B > $O/bench_ins_s512.json 2> $O/bench_ins_s512.err
timeout 900 python bench.py --insertion $B > $O/bench_ins_s1024.json 2> $O/bench_ins_s1024.err
timeout 900 python bench.py --insertion --streams 2 $B > $O/bench_ins_s1024_streams2.json 2> $O/bench_ins_s1024_streams2.err
timeout 1500 python bench.py --insertion --rollout-steps 800 --scenes 256 --insert-headroom 320 --steps 1 --warmup 1 $B --no-parity > $O/bench_c4shape_s256.json 2> $O/bench_c4shape_s256.err
timeout 900 python bench.py --insertion --rollout-steps 800 --scenes 128 --insert-headroom 320 --steps 2 --warmup 1 $B > $O/bench_c4shape_s128.json 2> $O/bench_c4shape_s128.err
timeout 900 python bench.py --agents 256 --map-tokens 4096 --rollout-steps 800 --scenes 32 --steps 2 --warmup 1 $B > $O/bench_c5shape_s32.json 2> $O/bench_c5shape_s32.err
timeout 300 python tools/bench_dropin.py 512 > $O/dropin.log 2>&1
for s in 1024 64 8; do timeout 300 python tools/soak_determinism.py $s 12 2>&1 | tail -1 >> $O/soak.log; done
bash tools/edge_by_set2.sh "" k_edge_fused3 > $O/edge_by_set.txt 2>&1
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']/1e6,3),'M', round(d['ms_per_step'],2),'ms frac', round(r['frac'],4), 'traffic_ratio', r.get('traffic_ratio'), d['config'].get('agents_inserted_last_rollout'))"; done
tail -1 $O/dropin.log; cat $O/soak.log
