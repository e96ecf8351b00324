"""micro-benchmark: k_fourier (fp32 MFMA) vs k_fourier_h (fp16 three-term split) on one edge set"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import make_weights
import os
from infgen_amd import _lib
if os.environ.get('EXP_LIB'):
    _lib.LIB_PATH = os.environ['EXP_LIB']
from infgen_amd import packing, engine
dev = torch.device('cuda:0')
lib = _lib.load()
ops = engine.Ops(dev)
sd = make_weights(seed=3)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 350000
for n, prefix in ((3, 'agent_encoder.r_a2a_emb'), (4, 'agent_encoder.r_t_emb')):
    pack = torch.from_numpy(packing.pack_fourier(sd, prefix, n)).to(dev)
    rng = np.random.default_rng(0)
    raw = np.zeros((E, 4), np.float32)
    raw[:, 0] = rng.uniform(0, 60, E); raw[:, 1:n] = rng.uniform(-np.pi, np.pi, (E, n - 1))
    rawd = torch.from_numpy(raw).to(dev)
    outs = {}
    for mode in (0, 1):
        _lib.check(lib.infgen_set_fourier_mode(mode))
        out = torch.empty(E, 128, device=dev)
        for _ in range(3):
            ops.fourier(rawd, n, pack, out, normalize=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.fourier(rawd, n, pack, out, normalize=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        flops = 2.0 * E * (n * 32896 + 16384)
        print(f'n={n} mode={mode} E={E}: {dt*1e6:.1f} us  {flops/dt/1e12:.1f} TFLOP/s (algorithmic)')
        outs[mode] = out.clone()
    print('   max |mode1 - mode0| =', float((outs[1] - outs[0]).abs().max()))
_lib.check(lib.infgen_set_fourier_mode(1))
# accuracy of both modes against an fp64 evaluation of the reference formula (fp32 sin/cos arguments)
from oracle import rollout_oracle as ro
tsd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
with torch.no_grad():
    x = torch.from_numpy(raw[:, :n])
    fr = torch.from_numpy(sd[prefix + '.freqs.weight'])
    z = (x.unsqueeze(-1) * fr * 2 * np.pi).double()           # fp32 arguments like the reference, exact sin/cos of them
    feats = torch.cat([z.cos(), z.sin(), x.double().unsqueeze(-1)], -1)
    acc = 0
    for i in range(n):
        h = ro._lin(tsd64, f'{prefix}.mlps.{i}.0', feats[:, i])
        h = torch.relu(ro._ln(tsd64, f'{prefix}.mlps.{i}.1', h))
        acc = acc + ro._lin(tsd64, f'{prefix}.mlps.{i}.3', h)
    o = ro._lin(tsd64, prefix + '.to_out.2', torch.relu(ro._ln(tsd64, prefix + '.to_out.0', acc)))
    ref = torch.nn.functional.layer_norm(o, (128,))
for m in (0, 1):
    err = (outs[m].cpu().double() - ref).abs().max(-1).values
    print('mode', m, 'max err', float(err.max()), 'rows > 1e-4:', int((err > 1e-4).sum()), ' > 1e-3:', int((err > 1e-3).sum()),
          'median', float(err.median()), 'worst idx', torch.topk(err, 6).indices.tolist())
