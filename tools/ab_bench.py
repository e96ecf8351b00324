"""same-box A/B of whole-rollout bench lines across library variants:
python tools/ab_bench.py [--scenes N] [--reps R] lib1 lib2 ...   ('' or 'shipped' = infgen_amd/libinfgen_hip.so; else a path under the repo)"""
import json, os, subprocess, sys
args = sys.argv[1:]
scenes, reps, extra = '512', 2, []
while args and args[0].startswith('--'):
    if args[0] == '--scenes': scenes = args[1]; args = args[2:]
    elif args[0] == '--reps': reps = int(args[1]); args = args[2:]
    elif args[0] == '--insertion': extra.append('--insertion'); args = args[1:]
    else: raise SystemExit('unknown flag ' + args[0])
libs = args or ['shipped']
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
code = '''
import os, sys
sys.path.insert(0, %r)
from infgen_amd import _lib
if os.environ.get("EXP_LIB_BENCH"): _lib.LIB_PATH = os.path.join(%r, os.environ["EXP_LIB_BENCH"])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-parity", "--no-literal", "--no-strict", "--scenes", %r, "--steps", "5"] + %r
import runpy; runpy.run_path(os.path.join(%r, "bench.py"), run_name="__main__")
''' % (root, root, scenes, extra, root)
for rep in range(reps):
    for lib in libs:
        env = dict(os.environ, EXP_LIB_BENCH='' if lib in ('', 'shipped') else lib)
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if line:
            d = json.loads(line[0]); pk = d['roofline']['per_kernel_ms_one_rollout']
            print(f"{lib or 'shipped':40s} scenes {scenes}: {d['value'] / 1e6:7.3f} M  {d['ms_per_step']:7.2f} ms ", {k: pk[k] for k in ('k_fourier', 'k_attn_post', 'k_edge_attn', 'k_heads', 'k_linear') if k in pk}, flush=True)
        else:
            print(lib, 'failed', out.stderr[-600:], flush=True)
