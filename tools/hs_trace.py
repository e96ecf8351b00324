"""phase timing of k_attn_hs (variant build -DIG_HS_TRACE=1): s_memtime of wave 0 of workgroup 0 along the chain.
EXP_LIB=build_exp/libinfgen_hip_hstrace.so python tools/hs_trace.py [rows] [has_pos] [with_u]"""
import sys, os, ctypes as C, numpy as np, torch
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import make_weights
from infgen_amd import _lib
_lib.LIB_PATH = os.path.join(R, os.environ['EXP_LIB'])
from infgen_amd import packing, engine
dev = torch.device('cuda:0'); lib = _lib.load(); ops = engine.Ops(dev); sd = make_weights(seed=3)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
has_pos = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with_u = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p1 = torch.from_numpy(packing.pack_attention_layer(sd, 'agent_encoder.t_attn_layers.0')).to(dev)
p2 = torch.from_numpy(packing.pack_attention_layer(sd, 'agent_encoder.pt2a_attn_layers.0')).to(dev)
g = torch.Generator(device='cpu').manual_seed(0)
f = lambda *s: torch.randn(*s, generator=g).to(dev)
X = f(rows, 128); AGG = f(rows, 128) * 0.5; Z = f(rows, 8, 128) * 0.3; SIG = torch.rand(rows, 8, generator=g).to(dev)
Q = torch.empty(rows, 128, device=dev); U = torch.empty(rows, 8, 128, device=dev)
K = torch.empty(rows, 128, device=dev); V = torch.empty(rows, 128, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.infgen_set_attn_mode(3))
names = ['start', 'tables staged', 'x/agg loaded + LN + frags', 'gate/self GEMMs', 'gate arithmetic', 'exchange', 'frags', 'out proj GEMM',
         'exchange', 'post norm', 'ffn pre norm + frags', 'FFN up (4 GEMMs)', 'barrier', 'FFN down (4 x read, frags, GEMM)', 'exchange', 'ffn post norm',
         'next pre norm + frags', 'q (+u)', 'k / v + end']
raw = C.CDLL(_lib.LIB_PATH)
import time
for rep in range(3):
    for _ in range(5):
        _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(), has_pos,
                                            p2.data_ptr(), Q.data_ptr(), U.data_ptr() if with_u else 0, K.data_ptr(), V.data_ptr(), st))
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert raw.infgen_debug_hs_trace(buf) == 0
    t = np.array(buf[:19], dtype=np.int64)
    e = np.array(buf[19:24], dtype=np.int64) - t[0]
    print('  prologue (cycles after start): pointers known', e[0], ' table loads issued', e[1], ' all first loads issued', e[2], ' table data arrived', e[3], ' rows arrived', e[4], ' barrier passed', t[1] - t[0])
    print(f'rows {rows} has_pos {has_pos}: total {t[18] - t[0]} cycles; ' + ', '.join(f'{names[i]} {t[i] - t[i - 1]}' for i in range(1, 19)))
t0 = time.perf_counter()
for _ in range(50):
    _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), Z.data_ptr(), SIG.data_ptr(), has_pos,
                                        p2.data_ptr(), Q.data_ptr(), U.data_ptr() if with_u else 0, K.data_ptr(), V.data_ptr(), st))
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch (back to back)')
