"""Inference twin of the fused model.

Reference: BIOINF_tesi/models/EmbraceNetMultimodal_NoTrain.py:94-214 -- the trained multimodal network rebuilt from
the checkpoint that Kfold_CV_Multimodal writes (``{cell_line}_EmbraceNetMultimodal[_augmentation]_{task}_{n_iter}_test_.pt``
holding ``model_state_dict`` and the best trial's ``model_params``, training_models_multimodal.py:639-641), with frozen
pre-networks and a softmax head whose output is flattened (:211-214).  ``visual.Compare_Models_Result``
(:266-293) then calls it once per region and keeps element [1], the probability of the positive class.

Here the class keeps that constructor / ``forward`` / state-dict surface (it IS the training-time module built from the
stored hyper-parameters, so the same HIP kernels run), and adds ``predict_proba`` -- the whole table in batches, one
fused forward per batch instead of 63 k-163 k single-row launches.
"""
import os
from collections import defaultdict

import torch

from .embracenet import EmbraceNetMultimodal
from .prenets import conv_output_length


class _StoredTrial:
    """Answers the constructors' ``trial.suggest_*`` calls from a stored ``model_params`` dictionary."""

    def __init__(self, params):
        self.params = dict(params)

    def _get(self, name):
        if name not in self.params:
            raise KeyError(f"checkpoint model_params lacks '{name}'")
        return self.params[name]

    def suggest_int(self, name, low, high):
        return int(self._get(name))

    def suggest_categorical(self, name, choices):
        return self._get(name)

    def suggest_float(self, name, low, high):
        return float(self._get(name))


def get_single_model_params(model_params, models=("CNN", "FFNN")):
    """utils/utils.py:360-375: {'CNN': {...}, 'FFNN': {...}} with the model prefix stripped from the keys."""
    out = defaultdict(dict)
    for m in ([models] if isinstance(models, str) else models):
        for k, v in model_params.items():
            if k.startswith(m):
                out[m][k[k.index("_") + 1:]] = v
    return out


def output_size_from_model_params(cnn_params):
    """utils/utils.py:178-202: flattened width of the sequence pre-network."""
    length, channels = 256, 4
    for i in range(int(cnn_params["n_layers"])):
        k = cnn_params[f"kernel_size_l{i}"]
        length = conv_output_length(conv_output_length(length, k, int((k - 1) / 2), 1), 10, 0, 2)
        channels = cnn_params[f"out_channels_l{i}"]
    return length * channels


def checkpoint_name(cell_line, task, n_iter, augmentation=False):
    tag = "EmbraceNetMultimodal_augmentation" if augmentation else "EmbraceNetMultimodal"
    return f"{cell_line}_{tag}_{task}_{n_iter}_test_.pt"


class EmbraceNetMultimodal_NoTrain(EmbraceNetMultimodal):
    def __init__(self, cell_line, task, n_iter, in_features_FFNN, device, augmentation=False, n_classes=2, args=None,
                 embracenet_dropout=True, checkpoint_dir=None):
        path = checkpoint_name(cell_line, task, n_iter, augmentation)
        if checkpoint_dir is not None:
            path = os.path.join(checkpoint_dir, path)
        # tensors and plain containers only: nothing in the file is executed
        state = torch.load(path, map_location=torch.device(device), weights_only=True)
        params = state["model_params"]
        super().__init__(_StoredTrial(params), cell_line, task, device, in_features_FFNN, n_classes=n_classes, args=args,
                         embracenet_dropout=embracenet_dropout)
        self.n_iter = n_iter
        self.model_params = dict(params)
        for p in list(self.FFNN.parameters()) + list(self.CNN.parameters()):      # :135-138
            p.requires_grad = False
        single = get_single_model_params(params)
        assert self.CNN_pre_output_size == output_size_from_model_params(single["CNN"])

    def forward(self, x, availabilities=None, selection_probabilities=None, is_training=False, embracenet_dropout=True):
        logits = super().forward(x, availabilities=availabilities, selection_probabilities=selection_probabilities,
                                 is_training=is_training, embracenet_dropout=embracenet_dropout)
        return torch.softmax(logits, dim=1).reshape(-1)             # nn.Softmax(dim=None) on [B, 2] is dim=1 (:211-214)

    @torch.no_grad()
    def predict_proba(self, x_ffnn, x_cnn, batch_size=4096):
        """Probability of the positive class for every row of the two tables (what get_model_predictions collects with one
        call per region): [N] tensor on the model's device.  Modality selection is sampled per element exactly as in
        ``forward`` (the reference samples at inference time too)."""
        was_training = self.training
        self.eval()
        dev = next(self.parameters()).device
        dt = self.compute_dtype or next(self.parameters()).dtype
        out = []
        try:
            for i in range(0, x_ffnn.shape[0], batch_size):
                a = x_ffnn[i:i + batch_size].to(dev, dtype=dt, non_blocking=True)
                b = x_cnn[i:i + batch_size].to(dev, dtype=dt, non_blocking=True)
                out.append(self.forward([a, b]).view(-1, self.n_classes)[:, 1])
        finally:
            self.train(was_training)
        return torch.cat(out) if out else torch.empty(0, device=dev)
