cd $GRAFT_REPO_ROOT
python tools/ab_bench.py --reps 2 shipped build_exp/libinfgen_hip_rsq.so
python tools/ab_bench.py --scenes 8 --reps 1 shipped build_exp/libinfgen_hip_rsq.so
cp infgen_amd/libinfgen_hip.so /tmp/keep.so; cp build_exp/libinfgen_hip_rsq.so infgen_amd/libinfgen_hip.so
python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -q 2>&1 | tail -4
cp /tmp/keep.so infgen_amd/libinfgen_hip.so
