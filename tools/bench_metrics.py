"""compute_distance_to_nearest_object on the device vs the CPU oracle (= the reference's torch
code).  One unit = one (evaluated object, step) cell, i.e. N box pairs.  Prints one JSON line."""
import json, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from infgen_amd.metrics import compute_distance_to_nearest_object
from oracle import metrics_oracle as mo
B, N, T, NE = 512, 64, 80, 64
rng = np.random.default_rng(1)
head = rng.uniform(-np.pi, np.pi, (B, N, 1)) + rng.uniform(-0.3, 0.3, (B, N, 1)) * (np.arange(T) * 0.1)
vel = rng.uniform(0, 10, (B, N, 1, 1)) * np.stack([np.cos(head), np.sin(head)], -1)
pos = rng.uniform(-60, 60, (B, N, 1, 2)) + np.cumsum(vel, 2) * 0.1
f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
cx, cy, hd = f(pos[..., 0]), f(pos[..., 1]), f(head)
ln, wd = f(rng.uniform(0.8, 5.5, (B, N, 1)) * np.ones((1, 1, T))), f(rng.uniform(0.5, 2.2, (B, N, 1)) * np.ones((1, 1, T)))
vt = torch.from_numpy(rng.random((B, N, T)) > 0.05)
mask = torch.ones(N, dtype=torch.bool)
dev = torch.device('cuda:0')
g = [a.to(dev) for a in (cx, cy, cx * 0, ln, wd, ln, hd, vt, mask)]
for _ in range(3): compute_distance_to_nearest_object(*g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): compute_distance_to_nearest_object(*g)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
cells = B * NE * T
pairs = cells * (N - 1)
torch.set_num_threads(16)
t0 = time.perf_counter()
nb = 4
for b in range(nb): mo.distance_to_nearest_object(cx[b], cy[b], ln[b], wd[b], hd[b], vt[b], mask)
dc = time.perf_counter() - t0
print(json.dumps({'metric': 'object-step cells / s (compute_distance_to_nearest_object)', 'value': cells / dt, 'unit': 'cells/s',
                  'scenes': B, 'objects': N, 'steps': T, 'ms_per_call': dt * 1e3, 'box_pairs_per_s': pairs / dt,
                  'roofline': {'bound': 'valu', 'achieved': pairs * 230.0 / dt / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s',
                               'frac': pairs * 230.0 / dt / 1e12 / 157.3, 'note': '~230 flops per box pair'},
                  'cpu_baseline': {'value': nb * NE * T / dc, 'unit': 'cells/s', 'cores': 16, 'kind': 'port',
                                   'sample': f'{nb} scenes, torch 16 threads, {dc:.2f} s'}}))
