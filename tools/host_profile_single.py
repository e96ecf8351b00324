import cProfile, pstats, os, sys, time, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from infgen_amd import synth
from test_boundary_cpu import _decoder
from test_modules_gpu import _load, _to_data
dev = torch.device('cuda:0')
cfg = synth.standard_config()
sd = synth.fill_state_dict(bench.load_shapes(), seed=1, rich=True)
scenes, vocab, map_vocab, grid = bench.build_scenes(cfg, range(2), 64, 1024)
dec = _decoder(cfg); _load(dec, sd); dec = dec.to(dev).eval()
datas = [_to_data(sc, dev) for sc in scenes]
for _ in range(3):
    dec.inference(dict(datas[0])); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = dec.inference(dict(datas[0])); torch.cuda.synchronize()
print('inference(1 scene):', round(1e3 * (time.perf_counter() - t0) / 5, 2), 'ms')
pr = cProfile.Profile(); pr.enable()
out = dec.inference(dict(datas[0])); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
