cd $GRAFT_REPO_ROOT
python tools/ab_bench.py --scenes 8 --reps 2 shipped build_exp/libinfgen_hip_w12.so
python tools/ab_bench.py --scenes 64 --reps 2 shipped build_exp/libinfgen_hip_w12.so
python tools/ab_bench.py --scenes 512 --reps 2 shipped build_exp/libinfgen_hip_w12.so
cp infgen_amd/libinfgen_hip.so /tmp/keep.so; cp build_exp/libinfgen_hip_w12.so infgen_amd/libinfgen_hip.so
python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -q 2>&1 | tail -3
cp /tmp/keep.so infgen_amd/libinfgen_hip.so
