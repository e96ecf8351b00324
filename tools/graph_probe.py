"""which part of a decode step survives HIP-graph capture + replay (diagnostics for RolloutEngine(use_graph=True))"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from infgen_amd import engine, synth, _lib
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(int(sys.argv[1]) if len(sys.argv) > 1 else 8), 64, 1024)
w = engine.PackedWeights(sd, cfg, dev)
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
eng.rollout(); torch.cuda.synchronize()
lib, ctx = eng.lib, C.byref(eng._ctx)
def probe(name, fn):
    eng.prologue(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn(torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        print(name, 'ok', flush=True)
    except Exception as e:
        print(name, 'FAILED', repr(e)[:200], flush=True)
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
tests = {
 'build_edges': lambda st: _lib.check(lib.infgen_build_edges(ctx, 1, 0, st)),
 'raw_feature': lambda st: _lib.check(lib.infgen_raw_feature(ctx, 2, st)),
 'integrate': lambda st: _lib.check(lib.infgen_integrate(ctx, 0, st)),
 'decode_layers': lambda st: _lib.check(lib.infgen_decode_layers(ctx, 1, 0, st)),
 'decode_step': lambda st: _lib.check(lib.infgen_decode_step(ctx, 0, st)),
 'rollout_run': lambda st: _lib.check(lib.infgen_rollout_run(ctx, 0, cfg.num_decode_steps, st)),
}
for k, f in tests.items():
    if which in ('all', k):
        probe(k, f)
