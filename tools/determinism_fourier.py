# self-consistency race detector: run the same launch N times, compare outputs bitwise
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import make_weights
import os
from infgen_amd import _lib
if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
from infgen_amd import packing, engine
dev = torch.device('cuda:0'); lib = _lib.load(); ops = engine.Ops(dev)
sd = make_weights(seed=3)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 350000
n, prefix = (4, 'agent_encoder.r_t_emb') if os.environ.get('N4') else (3, 'agent_encoder.r_a2a_emb')
pack = torch.from_numpy(packing.pack_fourier(sd, prefix, n)).to(dev)
rng = np.random.default_rng(0)
raw = np.zeros((E, 4), np.float32)
raw[:, 0] = rng.uniform(0, 60, E); raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (E, n - 1))
rawd = torch.from_numpy(raw).to(dev)
outs = []
for it in range(int(os.environ.get('REPS', '8'))):
    out = torch.empty(E, 128, device=dev); ops.fourier(rawd, n, pack, out, normalize=True)
    torch.cuda.synchronize(); outs.append(out)
ref = outs[0].view(torch.int32)
tot = 0
for o in outs[1:]:
    tot += int((o.view(torch.int32) != ref).any(-1).sum())
print('rows differing bitwise from run 0 over 7 reruns:', tot, ' non-finite rows in run 0:', int((~torch.isfinite(outs[0])).any(-1).sum()))
