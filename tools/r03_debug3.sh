cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
export PYTHONFAULTHANDLER=1
run() { echo "== $1"; shift; env "$@" timeout 300 python bench.py --scenes 64 --no-cpu-baseline --no-parity --steps 2 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-260; }
run "graph legs, default" X=1
run "graph legs, heads nosplit" INFGEN_HEADS_NOSPLIT=1
run "graph legs, fourier nomulti" INFGEN_FOURIER_NOMULTI=1
run "graph legs, both" INFGEN_HEADS_NOSPLIT=1 INFGEN_FOURIER_NOMULTI=1
run "graph off" INFGEN_GRAPH=0
echo "== hazard repro 2"; hipcc -O3 --offload-arch=gfx950 -I infgen_amd/csrc tools/hazard_repro2.hip -o /tmp/hazard_repro2 2>/dev/null && timeout 300 /tmp/hazard_repro2 200 96 2>&1 | tee gpurun_out/r03d/hazard_repro2.log | tail -5
