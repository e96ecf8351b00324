"""cProfile of the host side of InfGenDecoder.inference_batch (marshalling, engine construction, rollout, device epilogue).
python tools/host_profile_dropin.py [scenes]"""
import cProfile, pstats, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from infgen_amd import synth
from test_boundary_cpu import _decoder
from test_modules_gpu import _load, _to_data
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(S), 64, 1024)
dec = _decoder(cfg); _load(dec, sd); dec = dec.to(dev).eval()
datas = [_to_data(sc, dev) for sc in scenes]
for _ in range(2):
    dec.inference_batch([dict(d) for d in datas]); torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = dec.inference_batch([dict(d) for d in datas]); torch.cuda.synchronize()
pr.disable()
print('inference_batch', S, 'scenes:', round(1e3 * (time.perf_counter() - t0), 1), 'ms')
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
