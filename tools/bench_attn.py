"""micro-benchmark of the node-side attention kernel (post + fused pre of the next layer)"""
import sys, time, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import make_weights
import os
from infgen_amd import _lib
if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
from infgen_amd import packing, engine
dev = torch.device('cuda:0'); lib = _lib.load(); ops = engine.Ops(dev)
sd = make_weights(seed=3)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p1 = torch.from_numpy(packing.pack_attention_layer(sd, 'agent_encoder.t_attn_layers.0')).to(dev)
p2 = torch.from_numpy(packing.pack_attention_layer(sd, 'agent_encoder.pt2a_attn_layers.0')).to(dev)
g = torch.Generator(device='cpu').manual_seed(0)
f = lambda *s: torch.randn(*s, generator=g).to(dev)
X0 = f(rows, 128); AGG = f(rows, 128) * 0.5; Z = f(rows, 8, 128) * 0.3; SIG = torch.rand(rows, 8, generator=g).to(dev)
Q = torch.empty(rows, 128, device=dev); U = torch.empty(rows, 8, 128, device=dev)
K = torch.empty(rows, 128, device=dev); V = torch.empty(rows, 128, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
HAS_POS = int(os.environ.get('HAS_POS', '1'))
for mode in (0, 1, 3):
    _lib.check(lib.infgen_set_attn_mode(mode))
    def run():
        _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(), HAS_POS,
                                            p2.data_ptr(), Q.data_ptr(), U.data_ptr(), K.data_ptr(), V.data_ptr(), st))
    X = X0.clone(); run(); torch.cuda.synchronize()
    res[mode] = (X.clone(), Q.clone(), U.clone(), K.clone(), V.clone())
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fl = 2.0 * rows * 16384 * 17
    print(f'rows={rows} mode={mode}: {dt*1e6:.1f} us  {fl/dt/1e12:.1f} TFLOP/s (algorithmic)')
for n, a, b, c in zip('XQUKV', res[0], res[1], res[3]):
    print('  max |split - fp32mfma|', n, float((a - b).abs().max()), ' |16-row split - split|', float((c - b).abs().max()))
_lib.check(lib.infgen_set_attn_mode(2))
