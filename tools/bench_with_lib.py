"""bench.py with the library swapped for a variant build: EXP_LIB_BENCH=build_exp/libinfgen_hip_<name>.so python tools/bench_with_lib.py <bench flags>
(what tools/ab_bench.py does per line; used under rocprofv3 by tools/edge_by_set.sh)"""
import os, runpy, sys
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root)
from infgen_amd import _lib
if os.environ.get('EXP_LIB_BENCH'):
    _lib.LIB_PATH = os.path.join(root, os.environ['EXP_LIB_BENCH'])
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path(os.path.join(root, 'bench.py'), run_name='__main__')
