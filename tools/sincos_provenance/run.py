"""Where do the reference's CPU sin / cos values come from?  (bit-exact pre-processing needs the device to
reproduce them.)  Build container only.  Compares, on 160 k arguments in [-8, 8]:
  torch.sin / torch.cos (what the reference's preprocess.py / infgen.py call on CPU tensors)
  Sleef_sinf16_u10 / Sleef_sinf8_u10avx2 called directly in libtorch_cpu.so (call_sleef.c)
  a scalar C restatement of Sleef's u10 fast path (sleef_u10_restatement.c: constants and operation order read off the shipped code)
  the correctly rounded value (float64 libm, rounded once)
Result on this image (torch 2.10.0+rocm7.0, MKL enabled): the restatement equals both Sleef builds bit for bit, but torch.sin differs
from Sleef in 2.3 % of the arguments and from the correctly rounded value in 4.9 % - ATen sends float32 sin / cos through Intel MKL's
VML (vsSin / vsCos, aten/src/ATen/cpu/vml.h) whenever MKL is compiled in, for every tensor layout.  MKL is closed source: its
rounding cannot be restated, and a reference installation without MKL (or on a non-x86 host) produces other bits again.  The index
work downstream of these values (token matching, grid cells) therefore has no platform-independent bit pattern to be exact against."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
tmp = tempfile.mkdtemp()
subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', '-o', tmp + '/libsl.so', here + '/sleef_u10_restatement.c', '-lm'])
subprocess.check_call(['gcc', '-O1', '-mavx512f', '-mavx2', '-shared', '-fPIC', '-o', tmp + '/libcall.so', here + '/call_sleef.c', '-ldl'])
S, Cc = ctypes.CDLL(tmp + '/libsl.so'), ctypes.CDLL(tmp + '/libcall.so')
lib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libtorch_cpu.so')
assert Cc.init(lib.encode())
print('torch', torch.__version__, 'mkl', torch.backends.mkl.is_available(), 'cpu capability', torch.backends.cpu.get_cpu_capability())
x = np.random.default_rng(1).uniform(-8, 8, 160000).astype(np.float32)
ne = lambda a, b: int((a.view(np.uint32) != b.view(np.uint32)).sum())
for fn, tf, nf in (('sinf', torch.sin, np.sin), ('cosf', torch.cos, np.cos)):
    f = getattr(S, 'sl_' + fn); f.restype = ctypes.c_float; f.argtypes = [ctypes.c_float]
    mine = np.array([f(float(v)) for v in x], dtype=np.float32)
    t = tf(torch.from_numpy(x)).numpy()
    cr = nf(x.astype(np.float64)).astype(np.float32)
    row = [f'{fn}: torch vs restatement {ne(t, mine)}', f'torch vs correctly rounded {ne(t, cr)}', f'restatement vs correctly rounded {ne(mine, cr)}']
    for name, call in ((f'Sleef_{fn}16_u10', Cc.call16), (f'Sleef_{fn}8_u10avx2', Cc.call8)):
        out = np.zeros_like(x)
        if call(name.encode(), x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(x))) == 0:
            row.append(f'{name} vs restatement {ne(out, mine)} vs torch {ne(out, t)}')
    print(' | '.join(row), 'of', len(x))
