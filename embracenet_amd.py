"""Importable alias of the package directory (whose contract-mandated name contains hyphens).

    import embracenet_amd as ea;  ea.EmbraceNetMultimodal(...)

After this module is imported, ``sys.modules['embracenet_amd']`` *is* the real package, and its
submodules are reachable under both names without being loaded twice.
"""
import importlib
import os
import sys

_REAL = "prediction-of-active-and-inactive-regulatory-regions-with-embracenet-multimodal-neural-network-_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + "."):
        sys.modules["embracenet_amd" + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
