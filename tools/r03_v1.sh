cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py tests/test_modules_gpu.py -m gpu -x -q 2>&1 | tail -3
for sc in 8 64 512; do python tools/ab_bench.py --scenes $sc --reps 1 shipped; done
for c in 0 7 8; do echo -n "cut $c: "; INFGEN_HS_CUT=$c HAS_POS=0 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=3"; done
