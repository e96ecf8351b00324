"""teacher-forced forward: the device path (infgen_amd/forward_engine.py) timed next to the CPU oracle
(oracle/forward_oracle.py) on the committed two-scene batch (40 agents, 300 map tokens, 18 token columns: the batch of
tests/golden/forward_a40.npz).  python tools/bench_forward.py [reps] -> one JSON line"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from forward_case import load_forward_case
from infgen_amd import engine, forward_engine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
c = load_forward_case()
dev = torch.device('cuda:0')
w = engine.PackedWeights(c['sd'], c['cfg'], dev)
A, T = c['batch']['agent']['state_idx'].shape
M = int(c['batch']['pt_token']['num_nodes'])


def once():
    torch.manual_seed(c['meta']['rng_seed'])
    eng = forward_engine.ForwardEngine(w, c['batch'], c['vocab'], c['map_vocab'], c['grid'])
    out = eng.run()
    torch.cuda.synchronize()
    return eng, out


eng, out = once(); once()
t0 = time.perf_counter()
for _ in range(reps):
    once()
dt = (time.perf_counter() - t0) / reps
edges = {k: int(v) for k, v in eng.edge_counts.items()}
# the CPU oracle on the host (torch CPU, all host threads)
from oracle import forward_oracle as fo
tsd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
torch.manual_seed(c['meta']['rng_seed'])
t1 = time.perf_counter()
fo.run_forward(tsd, c['batch'], c['cfg'], c['vocab'], c['map_vocab'], c['grid'])
t_cpu = time.perf_counter() - t1
print(json.dumps(dict(what='teacher-forced forward, engine construction + run (edge build, 7 edge sets, 6 x 3 motion sublayers over every '
                           'column, seed coarse / refine stages, heads), two scenes', agents=int(A), columns=int(T), map_tokens=M,
                      edges=edges, ms_per_forward=1e3 * dt, agent_columns_per_s=A * T / dt,
                      cpu_oracle_s=t_cpu, cpu_threads=torch.get_num_threads(), speedup=t_cpu / dt)))
