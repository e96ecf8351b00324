cd $GRAFT_REPO_ROOT
echo "== ops tests (shipped)"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k fourier 2>&1 | tail -3
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_noap.so build_exp/libinfgen_hip_w12.so
NG=3 EXP_LIB=build_exp/libinfgen_hip_trace_w12.so python tools/fh_trace.py 400000 2>&1 | sed -n 20,27p
python -m pytest tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -3
