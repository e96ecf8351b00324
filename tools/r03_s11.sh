cd $GRAFT_REPO_ROOT
EXP_LIB=build_exp/libinfgen_hip_hstrace.so python tools/hs_trace.py 512 0 0 2>&1 | tail -2
EXP_LIB=build_exp/libinfgen_hip_hstrace.so python tools/hs_trace.py 4096 0 0 2>&1 | tail -1
for l in "" build_exp/libinfgen_hip_hsold.so; do echo "-- lib=$l"; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=3\|rror\|16-row"; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 4096 2>&1 | grep "mode=3\|rror"; done
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k attn 2>&1 | tail -3
python tools/ab_bench.py --scenes 8 --reps 2 shipped build_exp/libinfgen_hip_hsold.so
python tools/ab_bench.py --scenes 64 --reps 1 shipped build_exp/libinfgen_hip_hsold.so
