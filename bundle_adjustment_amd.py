"""Importable alias for the package directory ``bundle-adjustment_amd/`` (a hyphen is not a valid module name).

``import bundle_adjustment_amd`` and ``import bundle_adjustment_amd.x`` resolve to the very same module objects as
``bundle-adjustment_amd`` / ``bundle-adjustment_amd.x`` (no double import, so ctypes structure classes stay unique).
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "bundle-adjustment_amd"
_ALIAS = "bundle_adjustment_amd"
_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            real = _REAL + fullname[len(_ALIAS):]
            return importlib.util.spec_from_loader(fullname, _AliasLoader(real))
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[_ALIAS + _name[len(_REAL):]] = _mod
