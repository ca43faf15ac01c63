"""MI355X-native EmbraceNet fusion + classifier training path.

The directory name is fixed by the build contract and is not a Python identifier; import it as
``embracenet_amd`` (the one-file alias at the repository root) or through importlib.
Public surface mirrors the reference's BIOINF_tesi.models / BIOINF_tesi.models.utils for this path.
"""
from . import _lib, data, dist, functional, inference, metrics, optim, training  # noqa: F401
from .embracenet import EmbraceNet, EmbraceNetMultimodal  # noqa: F401
from .inference import EmbraceNetMultimodal_NoTrain  # noqa: F401
from .metrics import (AUPRC, EarlyStopping, F1_precision_recall, accuracy, get_input_size,  # noqa: F401
                      get_loss_weights_from_dataloader, get_loss_weights_from_labels, size_out_convolution,
                      weight_reset)
from .prenets import CNN_pre, FFNN_pre  # noqa: F401
from .training import Kfold_CV_Multimodal, Param_Search_Multimodal, fit_multimodal  # noqa: F401

__all__ = ["EmbraceNet", "EmbraceNetMultimodal", "EmbraceNetMultimodal_NoTrain", "FFNN_pre", "CNN_pre", "fit_multimodal", "Param_Search_Multimodal",
           "Kfold_CV_Multimodal", "EarlyStopping", "AUPRC", "F1_precision_recall", "get_loss_weights_from_labels",
           "weight_reset", "functional", "optim", "metrics", "training", "dist"]
