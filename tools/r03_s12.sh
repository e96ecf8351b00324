cd $GRAFT_REPO_ROOT
EXP_LIB=build_exp/libinfgen_hip_hstrace.so python tools/hs_trace.py 512 0 0 2>&1 | tail -2
INFGEN_ATTN_WARM=0 EXP_LIB=build_exp/libinfgen_hip_hstrace.so python tools/hs_trace.py 512 0 0 2>&1 | tail -2
for wm in 128 0; do echo "-- warm=$wm"; INFGEN_ATTN_WARM=$wm HAS_POS=0 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=3\|rror\|16-row"; INFGEN_ATTN_WARM=$wm HAS_POS=0 timeout 60 python tools/bench_attn.py 2048 2>&1 | grep "mode=3\|rror"; done
HAS_POS=0 timeout 60 python tools/bench_attn.py 32768 2>&1 | grep "mode=1\|rror"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
for wm in 128 0; do INFGEN_ATTN_WARM=$wm python tools/ab_bench.py --scenes 8 --reps 1 shipped; done
python tools/ab_bench.py --scenes 8 --reps 1 build_exp/libinfgen_hip_hsold.so
python tools/ab_bench.py --scenes 64 --reps 1 shipped build_exp/libinfgen_hip_hsold.so
python tools/ab_bench.py --scenes 512 --reps 1 shipped build_exp/libinfgen_hip_hsold.so
