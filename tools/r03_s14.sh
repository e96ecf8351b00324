cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_hsold.so; do echo "-- lib=$l"; for r in 512 2048 4096; do EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py $r 2>&1 | grep "mode=3\|rror\|16-row split - split| X"; done; done
EXP_LIB= HAS_POS=1 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=3\|rror\|16-row split - split"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k attn 2>&1 | tail -3
python tools/ab_bench.py --scenes 8 --reps 1 shipped build_exp/libinfgen_hip_hsold.so
python tools/ab_bench.py --scenes 64 --reps 1 shipped build_exp/libinfgen_hip_hsold.so
