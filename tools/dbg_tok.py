import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import GOLDEN
from infgen_amd import synth
from infgen_amd.modules import TokenProcessor
z = np.load(os.path.join(GOLDEN, 'tok_a48.npz'))
dev = torch.device('cuda:0')
v = synth.make_agent_vocab(2048)
tok3 = torch.stack([torch.from_numpy(v[k][:, -1]) for k in ('veh', 'ped', 'cyc')]).to(dev)
ty = torch.from_numpy(z['type']).to(dev)
idx, con, _ = TokenProcessor()._match_agent_token(torch.from_numpy(z['valid']).to(dev), torch.from_numpy(z['pos'][..., :2].copy()).to(dev),
    torch.from_numpy(z['heading']).to(dev), torch.from_numpy(z['shape']).to(dev), tok3, agent_type=ty)
idx = idx.cpu().numpy(); con = con.cpu().numpy()
d = np.abs(con - z['token_contour']).max((2, 3))
for a in range(48):
    bad = np.nonzero(d[a] > 0)[0]
    if len(bad):
        o = bad[0]
        print('agent', a, 'type', z['type'][a], 'first diff step', o, 'maxdiff', d[a, o], 'idx dev/ref', idx[a, o], z['token_index'][a, o],
              'valid pair', z['valid'][a, 5*o], z['valid'][a, 5*(o+1)], 'prev valid pair', (z['valid'][a, 5*(o-1)], z['valid'][a, 5*o]) if o else None)
        print('   dev', con[a, o].ravel()); print('   ref', z['token_contour'][a, o].ravel())
