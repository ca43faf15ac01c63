"""Input staging on the device: the data set lives in HBM, a batch is a row gather.

The reference feeds the harness two ``DataLoader`` objects per split (``Build_DataLoader_Pipeline.return_data``,
BIOINF_tesi/data_pipe/dataprepare.py:544-597): the training loader draws its batches from
``BalancePos_BatchSampler`` (:417-453 -- positives spread evenly over the batches), the test loader is a shuffled
``DataLoader(batch_size*2, generator=manual_seed(random_state+30))`` (:592-594); every row is fetched, converted and
moved to the device one sample at a time (``Dataset_Wrap.__getitem__``, :399-412).

Here the whole split is staged ONCE (an A549 split is ~30 MB against 288 GB of HBM): features in the compute dtype,
sequences as one byte per position (functional.pack_onehot) or as the loader's one-hot floats, labels as int64.  Per epoch
the host produces only the batch index lists -- the same lists the reference's samplers produce, pinned by fixture G11
and by a live comparison with torch's DataLoader -- uploads them in one copy, and every batch is ONE HIP gather launch
(emb_gather_rows: features, sequences and labels share the index list).  No per-sample Python, no per-batch host->device traffic.

    train = data.device_loaders(x1, seq, y, batch_size=1024, device="cuda", balanced=True)
    test  = data.device_loaders(x1_t, seq_t, y_t, batch_size=2048, device="cuda", balanced=False, random_state=123 + 30)
    fit_multimodal(model, train, test, ...)          # same {'FFNN': ..., 'CNN': ...} dictionaries as the reference
"""
import random

import numpy as np
import torch

from . import functional as F_


# ---------------------------------------------------------------------------------------------------------------------
# batch index lists (host; integer work, bit-exact with the reference's samplers)
class _PyShuffler:
    """random.Random(seed).shuffle on int64 numpy arrays: natively (emb_mt19937_shuffle restates the Mersenne twister and
    CPython's shuffle; ~1 ms for 130 k rows instead of ~50 ms in the interpreter) or, native=False, with `random` itself
    (the tests hold the two against each other and against fixture G11)."""

    def __init__(self, seed, native=True):
        self.rng = random.Random(seed)                          # == random.seed(seed) on the module generator
        self.native = native
        if native:
            from . import _lib
            st = self.rng.getstate()[1]
            self.state = np.array(st[:624], dtype=np.uint32)
            self.pos = np.array([st[624]], dtype=np.int32)
            self._fn, self._check = _lib.lib().emb_mt19937_shuffle, _lib.check

    def shuffle(self, items):
        """items: int64 numpy array, shuffled in place."""
        if not self.native:
            tmp = items.tolist()
            self.rng.shuffle(tmp)
            items[:] = tmp
            return
        self._check(self._fn(self.state.ctypes.data, self.pos.ctypes.data, items.ctypes.data, items.size), "emb_mt19937_shuffle")


class BalancedBatchSampler:
    """Index lists of dataprepare.py:417-453: both classes are shuffled with Python's Mersenne twister seeded with
    `random_state` (the lists stay shuffled from epoch to epoch, so epochs differ although the seed is re-applied), each
    is cut into n_batches+1 nearly equal chunks (numpy.array_split sizes), positive chunk i is joined with negative chunk
    n_batches-i, and the batches are shuffled.  ``len()`` is n_batches = ceil(n / batch_size) while n_batches+1 batches are
    produced, as in the reference (its harness divides by len(loader))."""

    def __init__(self, labels, batch_size, random_state=123, native=True):
        y = np.asarray(labels).reshape(-1)
        self.pos = np.flatnonzero(y == 1).astype(np.int64)
        self.neg = np.flatnonzero(y == 0).astype(np.int64)
        self.n, self.batch_size, self.random_state, self.native = int(y.size), int(batch_size), random_state, native
        self.n_batches = -(-self.n // self.batch_size)

    def __len__(self):
        return self.n_batches

    @staticmethod
    def _bounds(n, k):
        """numpy.array_split boundaries: the first n % k chunks hold one element more."""
        base, extra = divmod(n, k)
        sizes = np.full(k, base, dtype=np.int64)
        sizes[:extra] += 1
        return np.concatenate([[0], np.cumsum(sizes)])

    def epoch_arrays(self):
        """-> (flat int64 indices of the epoch, batch sizes); advances the sampler like one ``iter()`` of the reference."""
        sh = _PyShuffler(self.random_state, self.native)
        sh.shuffle(self.pos)
        sh.shuffle(self.neg)
        k = self.n_batches + 1
        order = np.arange(k, dtype=np.int64)
        sh.shuffle(order)                                        # shuffling the list of batches == permuting their order
        pb, nb = self._bounds(self.pos.size, k), self._bounds(self.neg.size, k)
        parts, sizes = [], []
        for b in order:
            q = k - 1 - b                                        # the negative chunks are paired in reverse
            parts += [self.pos[pb[b]:pb[b + 1]], self.neg[nb[q]:nb[q + 1]]]
            sizes.append(int(pb[b + 1] - pb[b] + nb[q + 1] - nb[q]))
        return (np.concatenate(parts) if parts else np.zeros(0, np.int64)), sizes

    def epoch(self):
        flat, sizes = self.epoch_arrays()
        out, at = [], 0
        for s_ in sizes:
            out.append(flat[at:at + s_].tolist())
            at += s_
        return out

    def __iter__(self):
        return iter(self.epoch())


class ShuffledBatchSampler:
    """Index lists of ``DataLoader(dataset, batch_size, shuffle=True, generator=g)`` (the reference's test loader,
    dataprepare.py:592-594): per epoch torch's DataLoader first draws a base seed from `g`, then its RandomSampler
    permutes; the last batch is kept short.  torch's own sampler classes do the permutation here, so the stream of `g`
    is consumed exactly as the DataLoader would (checked live in tests/test_host_and_abi.py)."""

    def __init__(self, n, batch_size, random_state=None, generator=None):
        self.n, self.batch_size = int(n), int(batch_size)
        self.generator = generator if generator is not None else torch.Generator("cpu")
        if generator is None:
            self.generator.manual_seed(0 if random_state is None else int(random_state))
        self._sampler = torch.utils.data.BatchSampler(
            torch.utils.data.RandomSampler(range(self.n), generator=self.generator), self.batch_size, drop_last=False)

    def __len__(self):
        return -(-self.n // self.batch_size)

    def epoch(self):
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)    # the DataLoader iterator's base seed
        return list(self._sampler)

    def __iter__(self):
        return iter(self.epoch())


# ---------------------------------------------------------------------------------------------------------------------
# device-resident split + gathered batches
class DeviceSplit:
    """One split staged in HBM.  x1: [N, F] features, seq: [N, 4, L] one-hot windows or [N, L] uint8 base codes,
    y: [N] / [N, 1] labels.  `feature_dtype` = the dtype batches are handed to the model in (the compute dtype);
    `pack_sequence` stores one byte per position (8x less HBM and gather traffic; CNN_pre accepts the codes)."""

    def __init__(self, x1, seq, y, device, feature_dtype=torch.float64, pack_sequence=True):
        dev = torch.device(device)
        x1, seq, y = torch.as_tensor(x1), torch.as_tensor(seq), torch.as_tensor(y)
        if not (len(x1) == len(seq) == len(y)):
            raise ValueError("x1, seq and y must have the same number of rows")
        self.n = len(x1)
        self.labels_host = y.reshape(-1).to(torch.int64).cpu().numpy()
        self.x1 = x1.to(dev, feature_dtype).contiguous()
        if seq.dim() == 3 and pack_sequence:
            seq = F_.pack_onehot(seq.to(dev))
        self.seq = (seq.to(dev) if seq.dtype == torch.uint8 else seq.to(dev, feature_dtype)).contiguous()
        self.y = y.reshape(-1).to(dev, torch.int64).contiguous()
        self.device = dev


class _Epochs:
    """The index lists of epoch e, shared by the two modality views (the reference builds two samplers with one seed;
    they walk through the same lists).  One upload per epoch."""

    def __init__(self, split, sampler):
        self.split, self.sampler = split, sampler
        self._made, self._next = {}, 0                           # epoch -> (flat device indices, [(offset, size)])
        self._gather = F_.RowGather((split.x1, split.seq, split.y))
        self._last = (None, None)                                # ((epoch, batch), gathered tensors)
        self._staging = {}                                       # batch size -> (x1, seq, y, y[:, None]) reused every batch

    def batch(self, e, i):
        """(x1, seq, y[:, None]) of batch i of epoch e: ONE gather launch serves both modality views (whichever asks first
        triggers it; the harness walks the two loaders in lockstep)."""
        if self._last[0] != (e, i):
            idx, spans = self.get(e)
            at, size = spans[i]
            bufs = self._staging.get(size)
            if bufs is None:                                     # one set of staging buffers per batch size: like a DataLoader
                x1, seq, y = self._gather(idx[at:at + size])     # with pinned buffers, a batch is valid until the next one;
                bufs = self._staging[size] = (x1, seq, y, y.view(-1, 1))
                for t in bufs:                                   # training.StepRunner may capture these addresses in its
                    t._emb_staging = True                        # step graphs instead of copying every batch
            else:
                self._gather(idx[at:at + size], out=bufs[:3])
            self._last = ((e, i), (bufs[0], bufs[1], bufs[3]))
        return self._last[1]

    def get(self, e):
        if e < self._next and e not in self._made:
            raise RuntimeError("the two modality loaders of a split must be iterated together (epoch already dropped)")
        while self._next <= e:                                   # epochs are produced in order, each exactly once
            self._made.pop(self._next - 2, None)
            self._make(self._next)
            self._next += 1
        return self._made[e]

    def _make(self, e):
        if hasattr(self.sampler, "epoch_arrays"):
            flat, sizes = self.sampler.epoch_arrays()
        else:
            batches = self.sampler.epoch()
            sizes = [len(b) for b in batches]
            flat = np.fromiter((i for b in batches for i in b), dtype=np.int64, count=sum(sizes))
        if flat.size and (flat.min() < 0 or flat.max() >= self.split.n):
            raise IndexError("batch sampler produced a row index outside the split")
        spans, at = [], 0
        for size in sizes:
            spans.append((at, size))
            at += size
        self._made[e] = (torch.from_numpy(flat).to(self.split.device), spans)


class _View:
    """What the harness sees as one DataLoader: iterating yields (data, target) device batches."""

    def __init__(self, epochs, which):
        self.epochs, self.which, self._epoch = epochs, which, 0

    def __len__(self):
        return len(self.epochs.sampler)

    @property
    def input_size(self):
        sp = self.epochs.split
        return (sp.x1 if self.which == "FFNN" else sp.seq).shape[1]

    def __iter__(self):
        e = self._epoch
        self._epoch += 1
        _, spans = self.epochs.get(e)
        pick = 0 if self.which == "FFNN" else 1
        for i, (_, size) in enumerate(spans):
            if size == 0:
                continue                                         # (array_split can leave an empty chunk in tiny splits)
            got = self.epochs.batch(e, i)
            yield got[pick], got[2]


def device_loaders(x1, seq, y, batch_size, device, balanced=True, random_state=123, feature_dtype=torch.float64,
                   pack_sequence=True):
    """-> {'FFNN': loader, 'CNN': loader} over a split staged on `device`; `balanced` picks the training sampler
    (BalancePos) or the shuffled test loader (the reference seeds that one with random_state + 30 and doubles the batch
    size itself, dataprepare.py:592-594 -- pass those values)."""
    split = x1 if isinstance(x1, DeviceSplit) else DeviceSplit(x1, seq, y, device, feature_dtype, pack_sequence)
    sampler = BalancedBatchSampler(split.labels_host, batch_size, random_state) if balanced else \
        ShuffledBatchSampler(split.n, batch_size, random_state)
    epochs = _Epochs(split, sampler)
    return {"FFNN": _View(epochs, "FFNN"), "CNN": _View(epochs, "CNN")}
