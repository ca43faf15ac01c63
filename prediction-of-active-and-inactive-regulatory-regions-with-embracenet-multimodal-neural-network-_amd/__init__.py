"""MI355X-native EmbraceNet fusion + classifier training path.

The directory name is fixed by the build contract and is not a Python identifier; import it as
``embracenet_amd`` (the one-file alias at the repository root) or through importlib.
Public surface mirrors the reference's BIOINF_tesi.models / BIOINF_tesi.models.utils for this path.
"""
from . import _lib, functional, optim  # noqa: F401
from .embracenet import EmbraceNet, EmbraceNetMultimodal  # noqa: F401
from .prenets import CNN_pre, FFNN_pre  # noqa: F401

__all__ = ["EmbraceNet", "EmbraceNetMultimodal", "FFNN_pre", "CNN_pre", "functional", "optim"]
