import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_case
from infgen_amd import engine, _lib
from oracle import rollout_oracle as ro
c = load_case('a24_m256_edge'); z = c['z']
dev = torch.device('cuda:0')
w = engine.PackedWeights(c['sd'], c['cfg'], dev)
eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True)
tsd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
ref = ro.run_scene(tsd, c['scene'], c['cfg'], c['vocab'], c['map_vocab'], c['grid'])
A = ref['pos_a'].shape[0]
# replicate prologue but stop before column-0 chain
import types
lib = eng.lib
orig = lib.infgen_decode_layers
eng.prologue()   # full prologue (X = raw col 1 now)
ctx = eng._ctx
st = eng.ops.stream
_lib.check(lib.infgen_raw_feature(C.byref(ctx), 0, st))
torch.cuda.synchronize()
print('raw col0 err per row', np.abs(eng.X.cpu().numpy()[:A] - ref['X'][0][:, 0].numpy()).max(-1).round(5))
print('raw2 col0', eng.raw2.cpu().numpy()[:4])
_lib.check(lib.infgen_build_edges(C.byref(ctx), 0, 1, st))
ops = eng.ops
rows = eng.rows
for i in range(6):
    torch.cuda.synchronize()
    print('layer', i, 'input err', np.abs(eng.X.cpu().numpy()[:A] - ref['X'][i][:, 0].numpy()).max(-1).round(5)[:6])
    for name, packs, kv in (('t', w.attn_t, True), ('m', w.attn_m, False), ('a', w.attn_a, True)):
        ed = eng.edges[name]
        if kv:
            ops.attn_pre(eng.X, packs[i], q=eng.Q, u=eng.U, k=eng.Ka, v=eng.Va)
            ops.edge_attn(rows, eng.Q, eng.U, eng.Ka, eng.Va, ed['off'], ed['cnt'], ed['src'], ed['rhat'], eng.AGG, eng.Z, eng.SIG)
        else:
            ops.attn_pre(eng.X, packs[i], q=eng.Q, u=eng.U)
            ops.edge_attn(rows, eng.Q, eng.U, eng.mapK[i], eng.mapV[i], ed['off'], ed['cnt'], ed['src'], ed['rhat'], eng.AGG, eng.Z, eng.SIG)
        ops.attn_post(eng.X, packs[i], eng.AGG, eng.Z, eng.SIG, True)
