"""Import alias: ``import qavit_amd`` -> the package in ``qa-vit_amd/`` (a directory name Python cannot import
by statement).  Sub-modules are reachable as ``qavit_amd.models`` etc."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("qa-vit_amd")
for _name, _mod in list(sys.modules.items()):
    if _name == "qa-vit_amd" or _name.startswith("qa-vit_amd."):
        sys.modules["qavit_amd" + _name[len("qa-vit_amd"):]] = _mod
sys.modules[__name__] = _pkg
