import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_case
from infgen_amd import engine, _lib
from oracle import insertion_oracle as io
name = sys.argv[1] if len(sys.argv) > 1 else 'ins_forced_a16_m256'
c = load_case(name); z = c['z']; m = c['meta']
cfg = c['cfg']; cfg.disable_insertion = False
tsd = {k: torch.from_numpy(v) for k, v in c['sd'].items()}
ref = io.run_scene_with_insertion(tsd, c['scene'], cfg, c['vocab'], c['map_vocab'], c['grid'], force_enter=(m['insertion']=='forced'))
print('oracle seed log', [(d['t'], d['enter'], d['cell'], d['occupied'], d['type']) for d in ref['seed_log']][:14])
dev = torch.device('cuda:0')
w = engine.PackedWeights(c['sd'], cfg, dev)
eng = engine.RolloutEngine(w, [c['scene']], c['vocab'], c['map_vocab'], c['grid'], store_logits=True, force_enter=(m['insertion']=='forced'))
eng.prologue()
# monkeypatch insert_decide to log
lib = eng.lib
orig = lib.infgen_insert_decide
log = []
for t in range(cfg.num_decode_steps):
    if t > 0:
        eng._insert_step(t)
    torch.cuda.synchronize()
    n = int(eng.n_agents[0].item())
    cells = [int(eng.gridtok[0, 1 + t, a].item()) for a in range(int(eng.ins['first_new'][0].item()), n)] if t > 0 else []
    print('t', t, 'n_agents', n, 'ref', int(z['n_agents_step'][t]), 'new cells', cells, 'types', eng.atype[0, :n].cpu().numpy()[int(eng.ins['first_new'][0].item()):] if t>0 else '')
    eng.step(t)
    torch.cuda.synchronize()
    lg = eng.logits[t, :n].cpu().numpy()
    nr = int(z['n_agents_step'][t])
    k = min(n, nr)
    print('    logits err', np.abs(lg[:k] - z['logits'][t, :k]).max())
    if n != nr: break
