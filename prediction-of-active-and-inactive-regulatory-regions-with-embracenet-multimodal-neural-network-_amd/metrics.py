"""Host-side helpers of the train/eval step -- same names and semantics as
BIOINF_tesi/models/utils/utils.py (EarlyStopping :23-67, AUPRC :80-86, F1_precision_recall :89-94,
get_loss_weights_from_labels :121-140, size_out_convolution :143-153, weight_reset :155-163,
get_input_size :165-175).

The per-batch metrics of the reference are functions of four integers only (TP, predicted-positive,
positive, n): sklearn's average_precision_score on HARD 0/1 predictions and the macro
precision/recall/F1 both have closed forms (pinned against sklearn by fixture G6).  The HIP loss kernel
counts those integers on the device, so a step needs no device->host copy; the scores are evaluated here
once per epoch from the count table.
"""
import numpy as np
import torch
import torch.nn as nn


class EarlyStopping:
    """utils/utils.py:23-67 -- stop after `patience` epochs without an improvement of `delta`."""

    def __init__(self, patience=4, verbose=False, delta=0, trace_func=print):
        self.patience, self.verbose, self.delta, self.trace_func = patience, verbose, delta, trace_func
        self.counter, self.best_score, self.early_stop = 0, None, False

    def __call__(self, score):
        if self.best_score is None:
            self.best_score = score
        elif score < self.best_score + self.delta:
            self.counter += 1
            self.trace_func(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_score, self.counter = score, 0


def confusion_counts(output, target):
    """(TP, predicted-positive, positive, n) of argmax predictions."""
    pred = torch.argmax(output, dim=1).reshape(-1)
    t = target.reshape(-1)
    return (int(((pred == 1) & (t == 1)).sum()), int((pred == 1).sum()), int((t == 1).sum()), int(t.numel()))


def ap_from_counts(tp, pp, p, n):
    """average_precision_score(target, hard_pred) (utils/utils.py:84-86), NaN -> 0."""
    if p == 0:
        return 0.0
    if pp == 0 or pp == n:
        return p / n
    r = tp / p
    return r * (tp / pp) + (1.0 - r) * (p / n)


def prf_from_counts(tp, pp, p, n):
    """precision_recall_fscore_support(target, pred, average='macro', zero_division=0)[:3] (utils/utils.py:94):
    mean over the labels present in target or prediction."""
    fn, fp = p - tp, pp - tp
    tn = n - tp - fn - fp
    rows = []
    for t_, f_p, f_n, present in ((tp, fp, fn, p > 0 or pp > 0), (tn, fn, fp, n - p > 0 or n - pp > 0)):
        if not present:
            continue
        prec = t_ / (t_ + f_p) if t_ + f_p > 0 else 0.0
        rec = t_ / (t_ + f_n) if t_ + f_n > 0 else 0.0
        f1 = 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0
        rows.append((prec, rec, f1))
    return np.array(rows).mean(0)


def AUPRC(output, target):
    return ap_from_counts(*confusion_counts(output, target))


def F1_precision_recall(output, target):
    return prf_from_counts(*confusion_counts(output, target))


def accuracy(output, target):
    return (torch.argmax(output, dim=1) == target).float().mean()


def get_loss_weights_from_labels(label):
    """utils/utils.py:121-140: inverse-number-of-samples weights (w_pos, w_neg) of one batch."""
    label = torch.as_tensor(np.asarray(label)) if not torch.is_tensor(label) else label
    pos = int((label == 1).sum())
    neg = int((label == 0).sum())
    pos_inv = 1 / pos if pos != 0 else 0
    neg_inv = 1 / neg if neg != 0 else 0
    return pos_inv / (neg_inv + pos_inv), neg_inv / (neg_inv + pos_inv)


def get_loss_weights_from_dataloader(dataloader):
    pos = tot = 0
    for _, j in dataloader:
        pos += int(j.sum())
        tot += len(j)
    neg = tot - pos
    pos_inv = 1 / pos if pos != 0 else 0
    neg_inv = 1 / neg if neg != 0 else 0
    return pos_inv / (neg_inv + pos_inv), neg_inv / (neg_inv + pos_inv)


def size_out_convolution(input_size, kernel, padding, stride):
    return int(((input_size + 2 * padding - kernel) / stride) + 1)


def weight_reset(x):
    """utils/utils.py:155-163: re-initialise Linear / Conv1d / LSTM (BatchNorm statistics survive)."""
    if isinstance(x, (nn.Conv1d, nn.Linear, nn.LSTM)):
        x.reset_parameters()


def get_input_size(data_loader):
    if hasattr(data_loader, "input_size"):                     # data.device_loaders: no epoch is consumed to find out
        return data_loader.input_size
    for d, _ in data_loader:
        return d.shape[1]


class StepTable:
    """Device-resident per-step loss and confusion counts; one device->host copy per epoch."""

    def __init__(self, capacity, device):
        self.capacity = int(capacity)
        self.loss = torch.zeros(self.capacity, dtype=torch.float32, device=device)
        self.counts = torch.zeros(self.capacity, 4, dtype=torch.int64, device=device)
        self.n = 0

    def slot(self):
        if self.n >= self.capacity:
            raise RuntimeError("StepTable full")
        i = self.n
        self.n += 1
        return self.loss[i:i + 1], self.counts[i]

    def fetch(self):
        """-> (losses [n], counts [n,4]) on the host; resets the table."""
        n, self.n = self.n, 0
        return self.loss[:n].cpu().numpy().astype(np.float64), self.counts[:n].cpu().numpy()
