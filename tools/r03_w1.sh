cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/ab_bench.py --scenes 8 --reps 2 shipped build_exp/libinfgen_hip_prev.so
python tools/ab_bench.py --scenes 64 --reps 2 shipped build_exp/libinfgen_hip_prev.so
