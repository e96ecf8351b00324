cd $GRAFT_REPO_ROOT
for l in "" build_exp/libinfgen_hip_qsu.so; do echo "-- lib=$l"; for r in 32768 16384; do EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py $r 2>&1 | grep "mode=1\|rror"; done; EXP_LIB=$l INFGEN_ATTN_WAVES=8 HAS_POS=0 timeout 60 python tools/bench_attn.py 32768 2>&1 | grep "mode=1" | sed 's/^/  (8 waves) /'; done
python tools/ab_bench.py --reps 1 shipped build_exp/libinfgen_hip_qsu.so
