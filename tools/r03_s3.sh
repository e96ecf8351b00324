cd $GRAFT_REPO_ROOT
echo "== fourier kernel: anti-phase (shipped) vs in-step (noap)"
for l in "" build_exp/libinfgen_hip_noap.so; do echo "-- lib=$l"; EXP_LIB=$l timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode1 - mode0\|mode 1 max err\|rror" | head -8; done
echo "== ops tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fourier" 2>&1 | tail -3
EXP_LIB=build_exp/libinfgen_hip_trace.so python tools/fh_trace.py 400000 2>&1 | sed -n 1,32p
