cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/ab_bench.py --scenes 512 --reps 3 shipped build_exp/libinfgen_hip_prev.so
