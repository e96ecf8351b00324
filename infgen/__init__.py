"""``infgen`` import surface over ``infgen_amd`` (missing 6): the reference's entry scripts import
``infgen.model.infgen.InfGen``, ``infgen.modules.*``, ``infgen.utils.func`` ... (reference run.py:103-105, val.py); with this repo
on ``sys.path`` instead of the reference those imports resolve to the MI355X implementation - ``infgen.X`` IS the module
``infgen_amd.X`` (one module object, not a copy).  Nothing else lives here.  (tests/golden/make_golden*.py put /root/reference in
front of the path before importing ``infgen``: there the name means the reference itself.)"""
import importlib
import importlib.abc
import importlib.util
import sys

import infgen_amd

_PREFIX = __name__ + '.'


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = 'infgen_amd.' + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return importlib.import_module('infgen_amd.' + spec.name[len(_PREFIX):])

    def exec_module(self, module):       # (already executed under its own name)
        pass


sys.meta_path.insert(0, _AliasFinder())
__path__ = list(infgen_amd.__path__)
