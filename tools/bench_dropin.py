"""end-to-end time of the drop-in entries (InfGenDecoder.inference_batch / inference incl. the host-side epilogue) next to the
engine's rollout of the same scenes.  python tools/bench_dropin.py [scenes]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from infgen_amd import engine, synth
from test_boundary_cpu import _decoder
from test_modules_gpu import _load, _to_data
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(S), 64, 1024)
dec = _decoder(cfg); _load(dec, sd); dec = dec.to(dev).eval()
datas = [_to_data(sc, dev) for sc in scenes]
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
w = dec._weights()
eng = engine.RolloutEngine(w, scenes, vocab, map_vocab, grid)
t_roll = t(eng.rollout)
t_out = t(lambda: eng.outputs())
t_ctor = t(lambda: engine.RolloutEngine(w, scenes, vocab, map_vocab, grid), 2)
t_batch = t(lambda: dec.inference_batch([dict(d) for d in datas]), 2)
t_one = t(lambda: dec.inference(dict(datas[0])), 3)
print(f'scenes {S}: engine.rollout {1e3*t_roll:.1f} ms | engine.outputs {1e3*t_out:.1f} ms | RolloutEngine() {1e3*t_ctor:.1f} ms | '
      f'inference_batch {1e3*t_batch:.1f} ms ({S*64*80/t_batch/1e6:.2f} M agent-steps/s) | inference(1 scene) {1e3*t_one:.1f} ms')
