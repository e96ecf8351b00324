cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
echo "== determinism probe, 1 WG/CU (shipped)"; python tools/determinism_probe2.py 64 2>&1 | tail -7
echo "== determinism probe, two 8-wave workgroups per CU"; INFGEN_EDGE_WG2=1 python tools/determinism_probe2.py 64 2>&1 | tail -7
echo "== decoupled halves probe (k_edge_fused_p) with the real-pair broadcasts"; INFGEN_EDGE_P=1 python tools/determinism_probe2.py 64 2>&1 | tail -7
echo "== bench A/B"; for v in 0 1 0 1; do INFGEN_EDGE_WG2=$v python bench.py --no-cpu-baseline --no-parity --no-literal --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WG2=$v', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'us')"; done
echo "== tests"; python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -3
INFGEN_EDGE_WG2=1 python -m pytest tests/test_ops_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -3
echo "== ring variants (EXP_LIB)"
for l in "" build_exp/libinfgen_hip_ring4.so build_exp/libinfgen_hip_fh4.so; do echo "-- lib=$l"; EXP_LIB=$l HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"; EXP_LIB=$l python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|max err" | head -4; done
