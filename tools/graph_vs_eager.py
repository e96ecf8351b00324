"""few scenes: the decode steps replayed from a captured HIP graph (RolloutEngine(use_graph=True)) / the whole rollout as one graph
(use_graph='all') against eager launches.  python tools/graph_vs_eager.py [scenes ...]"""
import os, sys, time, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import bench
from infgen_amd import engine, synth
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
w = engine.PackedWeights(sd, cfg, dev)
for S in [int(x) for x in sys.argv[1:]] or [8, 64]:
    scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(S), 64, 1024)
    for mode in (False, True, 'all'):
        e = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, use_graph=mode)
        for _ in range(4):
            e.rollout()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            e.rollout()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'scenes {S} use_graph={mode!s:5s}: {1e3 * dt:7.3f} ms per rollout  {S * 64 * 80 / dt / 1e6:6.3f} M agent-steps/s', flush=True)
        del e
