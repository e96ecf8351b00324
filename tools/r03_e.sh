cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
L=gpurun_out/r03e/hazard_bisect.log; : > $L
hr() { name="$1"; shift; hipcc -O3 --offload-arch=gfx950 -I infgen_amd/csrc "$@" tools/hazard_repro2.hip -o /tmp/hr2 2>/dev/null && { echo "== $name" | tee -a $L; timeout 200 /tmp/hr2 100 96 2>&1 | tail -2 | tee -a $L; }; }
hr "as is"
hr "matrix half without MFMAs" -DMAT_NO_MFMA
hr "matrix half without split_pair" -DMAT_NO_SPLIT
hr "matrix half without LDS traffic" -DMAT_NO_LDS
hr "matrix half: global loads + VALU only" -DMAT_NO_MFMA -DMAT_NO_SPLIT -DMAT_NO_LDS
hr "no packed fp32 instructions in the whole kernel" -Xclang -target-feature -Xclang -packed-fp32-ops
export PYTHONFAULTHANDLER=1
for i in 1 2 3; do echo "== graph legs on a side stream, run $i"; BENCH_SIDE_STREAM=1 timeout 300 python bench.py --scenes 64 --no-cpu-baseline --no-parity --steps 2 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; done
echo "== attn kernel: barriers / free-running"; HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"; INFGEN_QS_DBG=1 HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"
INFGEN_ATTN_WAVES=8 HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"; INFGEN_ATTN_WAVES=8 INFGEN_QS_DBG=1 HAS_POS=0 python tools/bench_attn.py 32768 2>&1 | grep "mode=1"
echo "== fourier kernel: barriers / free-running"; python tools/bench_fourier.py 400000 2>&1 | tail -4; INFGEN_QS_DBG=1 python tools/bench_fourier.py 400000 2>&1 | tail -4
