cd $GRAFT_REPO_ROOT
for v in skipm skipv; do echo "== $v"; EXP_LIB=build_exp/libinfgen_hip_trace_$v.so python tools/fh_trace.py 400000 2>&1 | sed -n 2,18p; EXP_LIB=build_exp/libinfgen_hip_trace_$v.so timeout 120 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E" ; done
