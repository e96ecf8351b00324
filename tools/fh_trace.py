"""slot timing of the anti-phase k_fourier_h (variant build -DIG_FH_TRACE=1): s_memtime of waves 0 (group E) and 4 (group O) of
workgroup 0 before / after every slot barrier.  EXP_LIB=build_exp/libinfgen_hip_trace.so python tools/fh_trace.py [rows]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from conftest import make_weights
from infgen_amd import _lib
_lib.LIB_PATH = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), os.environ['EXP_LIB'])
from infgen_amd import packing, engine
dev = torch.device('cuda:0'); lib = _lib.load(); ops = engine.Ops(dev); sd = make_weights(seed=3)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
n, prefix = 3, 'agent_encoder.r_a2a_emb'
pack = torch.from_numpy(packing.pack_fourier(sd, prefix, n)).to(dev)
rng = np.random.default_rng(0); raw = np.zeros((E, 4), np.float32)
raw[:, 0] = rng.uniform(0, 60, E); raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (E, n - 1))
rawd = torch.from_numpy(raw).to(dev); out = torch.empty(E, 128, device=dev)
for _ in range(3): ops.fourier(rawd, n, pack, out, normalize=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
raw_lib = C.CDLL(_lib.LIB_PATH)
assert raw_lib.infgen_debug_fh_trace(buf) == 0
NG = int(os.environ.get('NG', '2'))
t = np.array(buf[:], dtype=np.int64).reshape(4, 64, 2)
print('slot | per wave group: work wait (cycles of s_memtime; work = arrival at the barrier - release of the previous one) | slot length')
for s_ in range(1, 46):
    cells = []
    for r in range(NG):
        cells.append(f'{t[r, s_, 0] - t[r, s_ - 1, 1]:7d} {t[r, s_, 1] - t[r, s_, 0]:6d}')
    print(f'{s_:4d} | ' + ' | '.join(cells) + f' | {t[0, s_, 1] - t[0, s_ - 1, 1]}')
st = np.array(buf[384:504], dtype=np.int64)
print('stamps of wave 0 (differences, cycles):', [int(st[i + 1] - st[i]) for i in range(0, 40)])
