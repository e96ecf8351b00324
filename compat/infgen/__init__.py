"""``infgen`` import surface over ``infgen_amd``: the reference's entry scripts import ``infgen.model.infgen.InfGen``,
``infgen.modules.*``, ``infgen.utils.func`` ... (reference run.py:103-105, val.py).  A maintainer who wants those imports to
resolve to the MI355X implementation puts THIS directory's parent (``<repo>/compat``) and the repository root on ``sys.path``
instead of the reference checkout; ``infgen.X`` then IS the module ``infgen_amd.X`` (one module object, not a copy).

The package deliberately does not live in the repository root: the reference's ``infgen/`` is a namespace package (no
``__init__.py``) and a regular package of the same name on ``sys.path`` always wins over it, whatever the path order - in the
root it shadowed the reference for the golden-vector generators (tests/golden/make_golden*.py), which must import the
reference itself.  Nothing else lives here."""
import importlib
import importlib.abc
import importlib.util
import sys

import infgen_amd

_PREFIX = __name__ + '.'


class _AliasFinder(importlib.abc.MetaPathFinder):
    """``infgen.X`` -> the module object of ``infgen_amd.X``, registered under both names.  The real module keeps its own
    ``__spec__`` / ``__package__`` (relative imports inside infgen_amd and ``importlib.reload`` keep working): the alias is
    entered into ``sys.modules`` here and the returned spec's loader hands that same object back without re-initialising
    its import attributes."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = 'infgen_amd.' + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        module = importlib.import_module(real)
        return importlib.util.spec_from_loader(fullname, _AliasLoader(module), is_package=hasattr(module, '__path__'))


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module
        self._spec = module.__spec__
        self._package = module.__package__

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        # importlib's _init_module_attrs has just overwritten __spec__ (and may have touched __package__ / __loader__) with
        # the alias spec: put the real ones back
        module.__spec__ = self._spec
        module.__package__ = self._package
        module.__loader__ = self._spec.loader if self._spec is not None else module.__loader__


sys.meta_path.insert(0, _AliasFinder())
__path__ = list(infgen_amd.__path__)
# Sub-packages this repository does not rebuild (the dataset module ``infgen.datasets.*``, visualisation, the
# WOMD proto readers) stay the reference's own: with INFGEN_REFERENCE_ROOT = a checkout of the reference, ``infgen.X`` for an X that
# infgen_amd lacks resolves to <root>/infgen/X through the ordinary path finder - so the reference's run.py / val.py import
# ``infgen.model.infgen.InfGen`` (this implementation), ``infgen.utils.func`` (this implementation) and
# ``infgen.datasets.scalable_dataset`` (theirs) side by side, unchanged.
import os as _os
_ref = _os.environ.get('INFGEN_REFERENCE_ROOT')
if _ref and _os.path.isdir(_os.path.join(_ref, 'infgen')):
    __path__.append(_os.path.join(_ref, 'infgen'))
