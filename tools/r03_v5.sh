cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_w12.so; do echo "-- lib=$l"; EXP_LIB=$l timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode 1 max err\|rror" | head -8; done
NG=3 EXP_LIB=build_exp/libinfgen_hip_trace_w12.so python tools/fh_trace.py 400000 2>&1 | sed -n 2,26p
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_w12.so
