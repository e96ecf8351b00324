cd $GRAFT_REPO_ROOT
echo "== hw sincos accuracy"; hipcc -O3 --offload-arch=gfx950 tools/hw_sincos_probe.hip -o /tmp/hwsc 2>/dev/null && /tmp/hwsc
for l in "" build_exp/libinfgen_hip_nopk.so; do echo "-- lib=$l"; EXP_LIB=$l HAS_POS=0 timeout 60 python tools/bench_attn.py 32768 2>&1 | grep "mode=1\|split - fp32mfma| X\|rror" ; EXP_LIB=$l timeout 60 python tools/bench_fourier.py 400000 2>&1 | grep "mode=1 E\|mode 1 max err\|rror" | head -6; done
python - <<'P'
import os, subprocess, json, sys
for lib in ('build_exp/libinfgen_hip_nopk.so', '', 'build_exp/libinfgen_hip_nopk.so', ''):
    env = dict(os.environ, EXP_LIB_BENCH=lib)
    out = subprocess.run([sys.executable, '-c', '''
import os, sys
sys.path.insert(0, "/root/repo")
from infgen_amd import _lib
if os.environ.get("EXP_LIB_BENCH"): _lib.LIB_PATH = os.path.join("/root/repo", os.environ["EXP_LIB_BENCH"])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-parity", "--no-literal", "--steps", "5"]
import runpy; runpy.run_path("/root/repo/bench.py", run_name="__main__")
'''], capture_output=True, text=True, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if line:
        d = json.loads(line[0]); pk = d['roofline']['per_kernel_ms_one_rollout']
        print(lib or 'shipped', round(d['value'] / 1e6, 3), 'M', round(d['ms_per_step'], 2), 'ms', d['roofline']['kernel'], round(d['roofline']['avg_launch_us'], 1), 'us', {k: pk[k] for k in ('k_fourier', 'k_attn_post', 'k_edge_attn', 'k_heads', 'k_linear')})
    else:
        print(lib, 'failed', out.stderr[-400:])
P
echo "== strict C4 / C5 tests"; python -m pytest tests/test_baseline_shapes_gpu.py -m gpu -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -15
