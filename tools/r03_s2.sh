cd $GRAFT_REPO_ROOT
echo "== fourier kernel: anti-phase (shipped) vs in-step (noap)"
for l in "" build_exp/libinfgen_hip_noap.so; do echo "-- lib=$l"; EXP_LIB=$l timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode1 - mode0\|mode 1 max err\|rror" | head -8; done
echo "== ops tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fourier" 2>&1 | tail -3
echo "== rollout A/B"; python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_noap.so
