cd $GRAFT_REPO_ROOT
for c in 1 2 3 4 5 6 7 8 9 10 0; do echo -n "cut $c: "; INFGEN_HS_CUT=$c HAS_POS=0 timeout 60 python tools/bench_attn.py 512 2>&1 | grep "mode=3"; done
