cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_noap.so build_exp/libinfgen_hip_old.so; do echo "-- lib=$l"; EXP_LIB=$l timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode 1 max err\|rror" | head -8; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 32768 2>&1 | grep "mode=1\|rror"; done
echo "== ops tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
EXP_LIB=build_exp/libinfgen_hip_trace.so python tools/fh_trace.py 400000 2>&1 | sed -n 2,18p
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_noap.so build_exp/libinfgen_hip_old.so
