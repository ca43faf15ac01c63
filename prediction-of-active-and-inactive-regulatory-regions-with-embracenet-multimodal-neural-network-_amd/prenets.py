"""Per-modality pre-networks feeding the EmbraceNet docking layers.

Interface mirrors the reference (same constructor arguments, same Optuna parameter names asked in the
same order, same ``output_size`` attribute and the same state-dict keys ``model.{3i}.*`` /
``CNN_model.{5i}.*``) so checkpoints and the harness keep working:
  FFNN_pre  <- BIOINF_tesi/models/FFNN_pre.py:10-49   epigenomic features  [B,F]     -> [B,d0]
  CNN_pre   <- BIOINF_tesi/models/CNN_pre.py:10-76    one-hot DNA window   [B,4,256] -> [B,d1]

The ``nn.Sequential`` members only HOLD the parameters (names, shapes, init, ``weight_reset``); on a ROCm
device ``forward`` runs the hand-written kernels: FFNN_pre = fused Linear+ReLU+Dropout GEMMs
(csrc/linear.hip), CNN_pre = conv-as-GEMM on a channels-last im2col view + fused BN/ReLU/MaxPool
(csrc/convblock.hip).  There is no stock-operator path in the product: the tests that compare against stock torch
operators call the ``nn.Sequential`` members themselves (tests/helpers.stock_prenets).
"""
import torch.nn as nn

from . import functional as F_

_FFNN_UNITS = ([32, 64, 128, 256], [16, 32, 64, 128], [4, 16, 32, 64], [4, 16, 32])
_CNN_CHANNELS = ([16, 32, 64], [32, 64, 96], [64, 96, 128, 256], [128, 256, 512])
# RNG layer ids (dropout streams, include/embrace_hip.h): post stack 0-3, CNN blocks 4-7, FFNN layers 8-11
_CNN_LAYER_ID0, _FFNN_LAYER_ID0 = 4, 8


def conv_output_length(length, kernel, padding, stride):
    """utils/utils.py:143-153"""
    return int(((length + 2 * padding - kernel) / stride) + 1)


class FFNN_pre(nn.Module):
    def __init__(self, trial, in_features, device, classes=2):
        super().__init__()
        self.trial, self.classes, self.device = trial, classes, device
        n_layers = trial.suggest_int("FFNN_n_layers", 1, 4)
        stack, width = [], in_features
        for i in range(n_layers):
            out = trial.suggest_categorical(f"FFNN_n_units_l{i}", _FFNN_UNITS[i])
            p = trial.suggest_categorical(f"FFNN_dropout_l{i}", [0.0, 0.2, 0.3, 0.4] if i < 2 else [0.0, 0.4, 0.5])
            stack += [nn.Linear(width, out), nn.ReLU(), nn.Dropout(p)]
            width = out
        self.output_size = width
        self.model = nn.Sequential(*stack)
        self.compute_dtype = None

    def forward(self, x, rng=None):
        T = self.compute_dtype or self.model[0].weight.dtype
        mods = list(self.model)
        layers = [(mods[i].weight, mods[i].bias, True, float(mods[i + 2].p) if self.training else 0.0, _FFNN_LAYER_ID0 + i // 3)
                  for i in range(0, len(mods), 3)]
        return F_.mlp(x, layers, rng=rng, compute_dtype=T)

    def prelaunch(self, x, rng=None):
        """Forward whose launch is parked to ride on the sequence CNN's first kernel (functional.mlp_prelaunch); returns a
        handle for `attach`, or None when the stack does not qualify (call the module then)."""
        if not x.is_cuda:
            return None
        T = self.compute_dtype or self.model[0].weight.dtype
        mods = list(self.model)
        layers = [(mods[i].weight, mods[i].bias, True, float(mods[i + 2].p) if self.training else 0.0, _FFNN_LAYER_ID0 + i // 3)
                  for i in range(0, len(mods), 3)]
        return F_.mlp_prelaunch(x, layers, rng=rng, compute_dtype=T)

    @staticmethod
    def attach(handle):
        return F_.mlp_attach(handle)


class CNN_pre(nn.Module):
    def __init__(self, trial, device):
        super().__init__()
        self.trial, self.device = trial, device
        n_layers = trial.suggest_int("CNN_n_layers", 1, 4)
        stack, channels, length = [], 4, 256
        for i in range(n_layers):
            out = trial.suggest_categorical(f"CNN_out_channels_l{i}", _CNN_CHANNELS[i])
            k = trial.suggest_categorical(f"CNN_kernel_size_l{i}", [5, 11, 15])
            pad = int((k - 1) / 2)
            stack += [nn.Conv1d(channels, out, kernel_size=k, stride=1, padding=pad), nn.BatchNorm1d(out), nn.ReLU(),
                      nn.MaxPool1d(kernel_size=10, stride=2)]
            p = trial.suggest_categorical(f"CNN_dropout_l{i}", [0, 0.2, 0.3, 0.4] if i < 1 else [0, 0.4, 0.5])
            stack.append(nn.Dropout(p))
            channels = out
            length = conv_output_length(conv_output_length(length, k, pad, 1), 10, 0, 2)
        self.output_size = channels * length
        self.CNN_model = nn.Sequential(*stack)
        self.compute_dtype = None
        # data parallelism: False = every rank normalises with the statistics of its own rows (throughput default, what
        # torch DDP does); True = BatchNorm statistics of the GLOBAL batch, i.e. the single-process result of the
        # reference (CNN_pre.py:41), at one small all-reduce per block and direction.
        self.sync_batchnorm = False

    def forward(self, x, rng=None):
        mods = list(self.CNN_model)
        layers = []
        for i in range(0, len(mods), 5):
            conv, bn, drop = mods[i], mods[i + 1], mods[i + 4]
            layers.append(dict(conv=conv, bn=bn, drop_p=float(drop.p), layer_id=_CNN_LAYER_ID0 + i // 5))
        T = self.compute_dtype or mods[0].weight.dtype
        bn_sync = None
        if getattr(self, "sync_batchnorm", False) and self.training:   # (attribute absent in older pickles)
            from . import dist as D
            bn_sync = D.allreduce_sum_ if D.collectives_on() else None
        return F_.conv_stack(x, layers, self.training, rng=rng, compute_dtype=T, bn_sync=bn_sync)
