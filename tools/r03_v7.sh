cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_prev.so build_exp/libinfgen_hip_nostore.so; do echo "-- lib=$l"; for r in 32768 16384; do EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py $r 2>&1 | grep "mode=1\|rror"; done; done
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k attn 2>&1 | tail -2
python tools/ab_bench.py --reps 2 shipped build_exp/libinfgen_hip_prev.so
