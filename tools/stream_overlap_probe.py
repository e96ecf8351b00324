"""do small dependent launches of two HIP streams overlap?  chains of k_attn_hs launches (512 rows) on 1 / 2 / 4 streams"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import make_weights
from infgen_amd import _lib, packing
dev = torch.device('cuda:0'); lib = _lib.load()
sd = make_weights(seed=3)
p1 = torch.from_numpy(packing.pack_attention_layer(sd, 'agent_encoder.t_attn_layers.0')).to(dev)
rows, n = 512, 200
def bufs():
    return [torch.randn(rows, 128, device=dev) for _ in range(2)] + [torch.empty(rows, 128, device=dev)]
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    B = [bufs() for _ in range(ns)]
    def run():
        for i in range(n):
            for st, (X, AGG, Q) in zip(streams, B):
                _lib.check(lib.infgen_attn_post_pre(X.data_ptr(), rows, p1.data_ptr(), AGG.data_ptr(), None, None, 0, p1.data_ptr(),
                                                    Q.data_ptr(), None, None, None, st.cuda_stream))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{ns} stream(s): {n} launches each, {1e3 * dt:.2f} ms total, {1e6 * dt / n:.1f} us per chain step')
