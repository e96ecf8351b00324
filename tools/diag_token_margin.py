import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_tokens_gpu as T
from infgen_amd.modules import TokenProcessor
from oracle import token_match_oracle as tm
rng = np.random.default_rng(99)
A, Tn = 4096, 91
atype = rng.integers(0, 3, size=A)
speed = rng.uniform(0.0, 14.0, size=A) * np.where(atype == 1, 0.15, 1.0)
yaw = rng.uniform(-0.5, 0.5, size=A)
t = np.arange(Tn) * 0.1
head = rng.uniform(-np.pi, np.pi, size=A)[:, None] + yaw[:, None] * t[None] + rng.normal(0, 0.01, size=(A, Tn))
vel = speed[:, None, None] * np.stack([np.cos(head), np.sin(head)], -1)
pos = rng.uniform(-80, 80, size=(A, 1, 2)) + np.cumsum(vel, 1) * 0.1 + rng.normal(0, 0.02, size=(A, Tn, 2))
valid = rng.random((A, Tn)) > 0.05
shape = np.array([[2.0, 4.8], [1.0, 2.0], [1.0, 1.0]], np.float32)[atype]
dev = torch.device('cuda:0')
tok3 = T._vocab_last(dev)
pos_t, head_t = torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(head.astype(np.float32))
ref_idx, ref_con, margin = tm.match_agent_token(torch.from_numpy(valid), pos_t, head_t, torch.from_numpy(shape), tok3.cpu()[torch.from_numpy(atype)], return_margin=True)
idx, con, _ = TokenProcessor()._match_agent_token(torch.from_numpy(valid).to(dev), pos_t.to(dev), head_t.to(dev), torch.from_numpy(shape).to(dev), tok3, agent_type=torch.from_numpy(atype).to(dev))
idx = idx.cpu()
eq = idx == ref_idx
bad = (~eq.all(1)).nonzero().flatten()
print('agents differing', len(bad))
for a in bad.tolist():
    o = int((~eq[a]).float().argmax())
    print('agent', a, 'first differing step', o, 'margin there', float(margin[a, o]), 'min margin of agent', float(margin[a].min()), 'contour diff before', float((con.cpu()[a, :o] - ref_con[a, :o]).abs().max()) if o else 0.0)
