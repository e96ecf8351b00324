cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_noap.so build_exp/libinfgen_hip_w12.so; do echo "-- lib=$l"; EXP_LIB=$l timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode 1 max err\|rror" | head -8; done
echo "== ops tests (w12)"; EXP_LIB_TEST=1 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k fourier 2>&1 | tail -3
NG=2 EXP_LIB=build_exp/libinfgen_hip_trace.so python tools/fh_trace.py 400000 2>&1 | sed -n 2,18p
NG=3 EXP_LIB=build_exp/libinfgen_hip_trace_w12.so python tools/fh_trace.py 400000 2>&1 | sed -n 2,26p
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_noap.so build_exp/libinfgen_hip_w12.so
