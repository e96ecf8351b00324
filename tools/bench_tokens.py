"""Agent tokenisation: k_match_tokens vs the CPU oracle (= the reference's torch code).
One unit = one agent (18 steps x 2048 tokens x 4 corners).  Prints one JSON line."""
import json, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from infgen_amd import synth
from infgen_amd.modules import TokenProcessor
from oracle import token_match_oracle as tm
A = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 91
rng = np.random.default_rng(5)
atype = rng.integers(0, 3, size=A)
speed = rng.uniform(0.0, 14.0, size=A) * np.where(atype == 1, 0.15, 1.0)
t = np.arange(T) * 0.1
head = (rng.uniform(-np.pi, np.pi, size=A)[:, None] + rng.uniform(-0.5, 0.5, size=A)[:, None] * t[None]).astype(np.float32)
vel = speed[:, None, None] * np.stack([np.cos(head), np.sin(head)], -1)
pos = (rng.uniform(-80, 80, size=(A, 1, 2)) + np.cumsum(vel, 1) * 0.1).astype(np.float32)
valid = rng.random((A, T)) > 0.05
shape = np.array([[2.0, 4.8], [1.0, 2.0], [1.0, 1.0]], np.float32)[atype]
dev = torch.device('cuda:0')
v = synth.make_agent_vocab(2048)
tok3 = torch.stack([torch.from_numpy(v[k][:, -1]) for k in ('veh', 'ped', 'cyc')]).to(dev)
args = (torch.from_numpy(valid).to(dev), torch.from_numpy(pos).to(dev), torch.from_numpy(head).to(dev), torch.from_numpy(shape).to(dev))
ty = torch.from_numpy(atype).to(dev)
tp = TokenProcessor()
for _ in range(3): tp._match_agent_token(*args, tok3, agent_type=ty)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tp._match_agent_token(*args, tok3, agent_type=ty)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
# 18 steps x 2048 tokens x 4 corners x (2 fma + 2 mul + 2 add + 2 sub + mul + fma + sqrt + add = 12 ops, fma = 2 flops -> 15)
flops = A * 18 * 2048 * 4 * 15.0
n_cpu = 256
torch.set_num_threads(16)
c = (torch.from_numpy(valid[:n_cpu]), torch.from_numpy(pos[:n_cpu]), torch.from_numpy(head[:n_cpu]), torch.from_numpy(shape[:n_cpu]),
     tok3.cpu()[torch.from_numpy(atype[:n_cpu])])
t0 = time.perf_counter(); tm.match_agent_token(*c); dc = time.perf_counter() - t0
print(json.dumps({'metric': 'agents tokenised / s (TokenProcessor._match_agent_token)', 'value': A / dt, 'unit': 'agents/s',
                  'agents': A, 'ms_per_call': dt * 1e3,
                  'roofline': {'bound': 'valu', 'achieved': flops / dt / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s',
                               'frac': flops / dt / 1e12 / 157.3},
                  'cpu_baseline': {'value': n_cpu / dc, 'unit': 'agents/s', 'cores': 16, 'kind': 'port',
                                   'sample': f'{n_cpu} agents, torch 16 threads, {dc:.2f} s'}}))
