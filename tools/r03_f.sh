cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
L=gpurun_out/r03f/hazard_bisect2.log; : > $L
hr() { name="$1"; shift; hipcc -O3 --offload-arch=gfx950 -I infgen_amd/csrc "$@" tools/hazard_repro2.hip -o /tmp/hr2 2>/dev/null && { echo "== $name" | tee -a $L; timeout 200 /tmp/hr2 150 96 2>&1 | tail -1 | tee -a $L; }; }
hr "as is"
hr "SGPR broadcasts as fully written SGPR pairs (no op_sel on them)" -DIG_EDGE_SPAIR
hr "SGPR broadcasts moved to VGPR pairs" -DIG_EDGE_SVPAIR
hr "per-lane broadcasts as fully written VGPR pairs; SGPR forms unchanged" -DIG_EDGE_VPAIR
hr "both: no op_sel broadcast left in the loop" -DIG_EDGE_SPAIR -DIG_EDGE_VPAIR
hr "as is, again" 
