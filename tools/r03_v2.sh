cd $GRAFT_REPO_ROOT
for w in 1 0 1 0; do echo -n "INFGEN_WARM=$w "; INFGEN_WARM=$w python tools/ab_bench.py --scenes 8 --reps 1 shipped; done
for w in 1 0; do echo -n "INFGEN_WARM=$w "; INFGEN_WARM=$w python tools/ab_bench.py --scenes 4 --reps 1 shipped; done
for w in 1 0; do echo -n "INFGEN_WARM=$w "; INFGEN_WARM=$w python tools/ab_bench.py --scenes 64 --reps 1 shipped; done
python -m pytest tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -2
