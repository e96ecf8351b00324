cd $GRAFT_REPO_ROOT
for d in 0 2 1; do echo "== INFGEN_QS_DBG=$d"; INFGEN_QS_DBG=$d HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"; INFGEN_QS_DBG=$d python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E"; done
