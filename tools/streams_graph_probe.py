"""the batch split over several engines / HIP streams with the WHOLE rollout of each engine as one HIP graph (RolloutEngine
use_graph='all'): python tools/streams_graph_probe.py <scenes> <streams> [<streams> ...]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from infgen_amd import engine, synth
S = int(sys.argv[1]); NS = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(S), 64, 1024)
w = engine.PackedWeights(sd, cfg, dev)
ref = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid, store_logits=False, use_graph=False)
ref.rollout(); torch.cuda.synchronize()
tok_ref = ref.next_token_all.clone() if hasattr(ref, 'next_token_all') else None
out_ref = [o['next_token_idx'] for o in ref.outputs()]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
t1 = timeit(ref.rollout)
print(f'scenes {S}: eager, one stream: {1e3 * t1:.2f} ms = {S * 64 * 80 / t1 / 1e6:.2f} M agent-steps/s', flush=True)
for ns in NS:
    per = (S + ns - 1) // ns
    for mode in ('all',):
        engs = [engine.RolloutEngine(w, scenes[i * per:(i + 1) * per], vocab, map_vocab, grid, store_logits=False, use_graph=mode)
                for i in range(ns) if scenes[i * per:(i + 1) * per]]
        streams = [torch.cuda.Stream(device=dev) for _ in engs]
        fn = (lambda: engine.rollout_many(engs, streams)) if ns > 1 else engs[0].rollout
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t = timeit(fn)
        ok = True
        k = 0
        for e in engs:
            for o in e.outputs():
                ok &= bool(np.array_equal(o['next_token_idx'], out_ref[k])); k += 1
        print(f'scenes {S}: whole-rollout graphs, {ns} stream(s): {1e3 * t:.2f} ms = {S * 64 * 80 / t / 1e6:.2f} M agent-steps/s; tokens equal to the eager run: {ok}', flush=True)
        del engs
