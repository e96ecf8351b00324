"""The reference's entry scripts import against compat/infgen UNCHANGED (north_star: "run.py / val.py drop in").

Build container only (needs /root/reference): the import blocks and function definitions of the reference's ``run.py`` and
``val.py`` (everything above ``if __name__ == '__main__'``) are executed in a fresh interpreter whose path holds ``<repo>/compat``
and ``<repo>`` and whose environment names the reference checkout (INFGEN_REFERENCE_ROOT), with stand-ins for the third-party
packages this image lacks (Lightning, PyG ...: tests/golden/_standins.py, bind=False).  Then
  * ``infgen.model.infgen.InfGen`` and ``infgen.utils.func`` are THIS repository's modules (infgen_amd),
  * ``infgen.datasets.scalable_dataset`` - which this repository does not rebuild - is the reference's own file,
  * the names the scripts use from ``infgen.utils.func`` behave: ``load_config_act`` reads the reference's own YAML configs,
    ``RankedLogger`` / ``Logging`` log, ``CONSOLE`` prints.
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = '/root/reference'

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, 'run.py')),
                                reason='needs /root/reference (build container only)')

CODE = r'''
import ast, inspect, logging, os, sys
sys.path.insert(0, %(golden)r)
import _standins
_standins.install(bind=False)
for script in ('run.py', 'val.py'):
    tree = ast.parse(open(os.path.join(%(ref)r, script)).read())
    top = [n for n in tree.body if not (isinstance(n, ast.If) and '__main__' in ast.dump(n.test))]
    ns = {'__name__': 'entry_' + script[:-3]}
    exec(compile(ast.Module(top, []), script, 'exec'), ns)
    print('TOP', script, sorted(k for k in ns if k in ('InfGen', 'MultiDataModule', 'MultiDataset', 'load_config_act',
                                                         'RankedLogger', 'Logging', 'CONSOLE', 'backup')))
import infgen, infgen.model.infgen as M, infgen.utils.func as F, infgen.datasets.scalable_dataset as D
import infgen_amd.model.infgen, infgen_amd.utils.func
print('SAME_MODEL', M is infgen_amd.model.infgen, inspect.getsourcefile(M.InfGen))
print('SAME_FUNC', F is infgen_amd.utils.func)
print('DATA', D.__file__)
cfg = F.load_config_act(os.path.join(%(ref)r, 'configs', 'ours_standard.yaml'))
print('CFG', cfg.Model.hidden_dim, cfg.Model.num_heads, cfg.Model.decoder.num_future_steps, type(cfg.Model.decoder).__name__)
os.environ['INFGEN_LOG_DIR'] = %(logs)r
lg = F.Logging().log(level='DEBUG', name='compat-test')
lg.info('file and console handlers')
os.environ['RANK'] = '1'
rl = F.RankedLogger('compat-ranked', rank_zero_only=True)
rl.logger.setLevel(logging.INFO)
rl.logger.addHandler(logging.StreamHandler(sys.stdout))
rl.info('from rank one: dropped')
os.environ['RANK'] = '0'
rl.info('from rank zero: kept')
F.CONSOLE.print('console ok')
print('LOGFILES', len(os.listdir(%(logs)r)))
'''


def test_reference_entry_scripts_import_against_the_alias_package(tmp_path):
    logs = str(tmp_path / 'logs')
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(REPO, 'compat'), REPO])
    env['INFGEN_REFERENCE_ROOT'] = REFERENCE
    env['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
    code = CODE % dict(golden=os.path.join(HERE, 'golden'), ref=REFERENCE, logs=logs)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    get = lambda tag: [ln for ln in lines if ln.startswith(tag + ' ')][-1]
    assert "TOP run.py ['CONSOLE', 'InfGen', 'MultiDataModule', 'RankedLogger', 'backup', 'load_config_act']" in out.stdout
    assert "TOP val.py ['CONSOLE', 'InfGen', 'Logging', 'MultiDataset', 'load_config_act']" in out.stdout
    assert get('SAME_MODEL').split()[1] == 'True' and os.path.join(REPO, 'infgen_amd', 'model', 'infgen.py') in get('SAME_MODEL')
    assert get('SAME_FUNC').split()[1] == 'True'
    assert get('DATA').split()[1] == os.path.join(REFERENCE, 'infgen', 'datasets', 'scalable_dataset.py')
    assert get('CFG') == 'CFG 128 8 80 ConfigDict'
    assert 'from rank zero: kept' in out.stdout and 'dropped' not in out.stdout
    assert 'console ok' in out.stdout and get('LOGFILES') == 'LOGFILES 1'


def test_config_dict_is_a_nested_attribute_dict():
    from infgen_amd.utils.func import ConfigDict
    c = ConfigDict({'a': {'b': [1, {'c': 2}]}, 'd': 3})
    assert c.a.b[1].c == 2 and c['a']['b'][0] == 1 and c.d == 3
    c.e = {'f': 4}
    assert c.e.f == 4 and isinstance(c.e, ConfigDict)
    with pytest.raises(AttributeError):
        c.missing
