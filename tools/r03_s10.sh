cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_gq0.so; do echo "-- lib=$l"; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 32768 2>&1 | grep "mode=\|rror"; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=\|rror";  done
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_gq0.so
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
