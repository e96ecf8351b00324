"""The golden recipe runs at HEAD and reproduces the committed fixtures from the REFERENCE's own code.

Build container only: skipped where /root/reference is absent (the GPU box).  Each test runs a generator of tests/golden/ in
a fresh interpreter with the repository root AND its compat/ alias directory on the path - the worst case: a regular package
called ``infgen`` is importable - and checks (1) that the classes the generator ran were defined under /root/reference
(``_standins.assert_reference`` inside the generators fails otherwise; a marker line is printed and checked here too) and
(2) that every array it wrote is bitwise equal to the committed fixture.
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, 'golden')
REFERENCE = '/root/reference'

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'infgen')),
                                reason='needs /root/reference (build container only)')


def _env():
    env = dict(os.environ)
    # the alias package importable on purpose: the generators must bind `infgen` to the reference anyway
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(REPO, 'compat'), REPO, env.get('PYTHONPATH', '')])
    env['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
    return env


def _same(path_new, path_old):
    a, b = np.load(path_new, allow_pickle=False), np.load(path_old, allow_pickle=False)
    assert sorted(a.files) == sorted(b.files), set(a.files) ^ set(b.files)
    for k in a.files:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == 'f'), k


def test_rollout_fixture_regenerates_bit_for_bit(tmp_path):
    """make_golden.py --cases c1_a8_m128: the reference's InfGenDecoder.inference (infgen/modules/infgen_decoder.py:123-130)"""
    code = ('import sys, runpy, inspect; sys.argv = ["make_golden.py", "--cases", "c1_a8_m128", "--out", %r]; '
            'runpy.run_path(%r, run_name="__main__"); '
            'from infgen.modules.infgen_decoder import InfGenDecoder as D; print("SRC", inspect.getsourcefile(D))'
            % (str(tmp_path), os.path.join(GOLDEN, 'make_golden.py')))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, env=_env(), cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    src = [ln for ln in out.stdout.splitlines() if ln.startswith('SRC ')][-1].split(' ', 1)[1]
    assert src.startswith(REFERENCE + '/'), src
    _same(os.path.join(str(tmp_path), 'c1_a8_m128.npz'), os.path.join(GOLDEN, 'c1_a8_m128.npz'))


def test_internals_fixture_regenerates_bit_for_bit(tmp_path):
    """make_golden_internals.py: edge lists and triple outputs hooked out of the reference's own agent_decoder.py:540-758, :2133-2158"""
    code = ('import sys, runpy; sys.argv = ["make_golden_internals.py", "--cases", "c1_a8_m128", "--out", %r]; '
            'runpy.run_path(%r, run_name="__main__")' % (str(tmp_path), os.path.join(GOLDEN, 'make_golden_internals.py')))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, env=_env(), cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    _same(os.path.join(str(tmp_path), 'c1_a8_m128_internals.npz'), os.path.join(GOLDEN, 'c1_a8_m128_internals.npz'))


@pytest.mark.parametrize('script, fixtures', [
    ('make_golden_tokenizer.py', ['attr_tokenizer.npz']),                # Attr_Tokenizer, attr_tokenizer.py:8-110
    ('make_golden_tokens.py', ['tok_a7.npz', 'maptok_p3.npz']),          # 8f rank 1: preprocess.py:552-653, infgen.py:918-936
    ('make_golden_metrics.py', ['dist_n5_t4.npz']),                      # 8f rank 2: interact_features.py:19-95
])
def test_widening_fixture_regenerates_bit_for_bit(tmp_path, script, fixtures):
    """the generators write next to themselves: run a copy of tests/golden/ in a scratch tree whose root is on the path"""
    root = tmp_path / 'tree'
    (root / 'tests').mkdir(parents=True)
    shutil.copytree(GOLDEN, root / 'tests' / 'golden', ignore=shutil.ignore_patterns('__pycache__'))
    for f in fixtures:
        os.remove(root / 'tests' / 'golden' / f)
    for name in ('infgen_amd', 'oracle'):
        os.symlink(os.path.join(REPO, name), root / name)
    out = subprocess.run([sys.executable, str(root / 'tests' / 'golden' / script)], capture_output=True, text=True, timeout=900,
                         env=_env(), cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    for f in fixtures:
        _same(str(root / 'tests' / 'golden' / f), os.path.join(GOLDEN, f))


def test_a_shadowing_package_is_refused(tmp_path):
    """`infgen` imported from anywhere else before the stand-ins are installed is an error, not a silent fixture of our own code"""
    code = ('import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); '
            'import infgen; import _standins\n'
            'try:\n    _standins.install()\nexcept RuntimeError as e:\n    print("REFUSED", e)\n'
            % (REPO, os.path.join(REPO, 'compat'), GOLDEN))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-1500:]
    assert 'REFUSED' in out.stdout and 'not from the reference' in out.stdout
