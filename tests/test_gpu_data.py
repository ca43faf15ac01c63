"""GPU: input staging -- the row-gather kernel against torch indexing (bit-exact: it only moves bytes),
and the device-resident loaders against the same batches fed from host lists."""
import numpy as np
import pytest
import torch

from oracle import datagen as dg
from oracle.configs import CONFIGS, FixedTrial

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n_idx", [0, 1, 37, 1024])
def test_gather_rows_equals_indexing(ea, n_idx):
    F = ea.functional
    g = torch.Generator(device="cpu").manual_seed(5)
    N = 301
    tables = [torch.rand(N, 48, generator=g).to(torch.bfloat16),              # 96-byte rows  (16-byte units)
              torch.randint(0, 5, (N, 256), generator=g, dtype=torch.uint8),   # 256-byte rows
              torch.randint(0, 2, (N,), generator=g, dtype=torch.int64),       # 8-byte rows
              torch.rand(N, 5, generator=g, dtype=torch.float64)]              # 40-byte rows (8-byte units)
    tables = [t.to(DEV) for t in tables]
    idx = torch.randint(0, N, (n_idx,), generator=g, dtype=torch.int64).to(DEV)
    outs = F.gather_rows(tables, idx)
    for t, o in zip(tables, outs):
        assert o.dtype == t.dtype and tuple(o.shape) == (n_idx,) + tuple(t.shape[1:])
        assert torch.equal(o, t[idx])


def test_gather_rows_odd_rows_views_and_bad_indices(ea):
    F = ea.functional
    g = torch.Generator(device="cpu").manual_seed(6)
    odd = torch.randint(0, 255, (50, 3), generator=g, dtype=torch.uint8).to(DEV)        # 3-byte rows: byte units
    f32 = torch.rand(50, 4, 7, generator=g).to(DEV)                                     # 112-byte rows, 3-D table
    flat = torch.randint(0, 50, (64,), generator=g, dtype=torch.int64).to(DEV)
    idx = flat[9:41]                                                                     # a view with a storage offset
    a, b = F.gather_rows((odd, f32), idx)
    assert torch.equal(a, odd[idx]) and torch.equal(b, f32[idx])
    bad = torch.tensor([3, -1, 50, 49], dtype=torch.int64, device=DEV)                   # out of range -> zero rows
    o, = F.gather_rows((f32,), bad)
    assert torch.equal(o[0], f32[3]) and torch.equal(o[3], f32[49]) and not o[1].any() and not o[2].any()
    with pytest.raises(TypeError):
        F.gather_rows((f32,), idx.to(torch.int32))
    with pytest.raises(Exception):
        F.gather_rows((f32.cpu(),), idx)


def _split(tag, n, F_in, rate):
    x1 = dg.uniform(f"{tag}/x1", (n, F_in))
    seq = dg.onehot_sequence(f"{tag}/seq", n)
    y = dg.labels(f"{tag}/y", n, rate)
    return x1, seq, y


@pytest.mark.parametrize("pack", [True, False])
def test_device_loaders_feed_the_harness_like_host_batches(ea, pack):
    """fit_multimodal over data.device_loaders (balanced training sampler, shuffled test loader; sequences as byte codes or
    one-hot floats) gives exactly the scores and parameters of fit_multimodal over host lists holding the same batches."""
    from embracenet_amd import data, optim, training
    hp, F_in = CONFIGS["small"]
    xtr, str_, ytr = _split("dl/train", 150, F_in, 0.3)
    xte, ste, yte = _split("dl/test", 70, F_in, 0.3)
    epochs, bs = 4, 32

    def host_lists(x1, seq, y, sampler):
        per_epoch = []
        for _ in range(epochs):
            ff, cc = [], []
            for b in sampler.epoch():
                if len(b):
                    t = torch.from_numpy(y[b])
                    ff.append((torch.from_numpy(x1[b]).float(), t)); cc.append((torch.from_numpy(seq[b]).float(), t))
            per_epoch.append((ff, cc))
        return per_epoch

    class Replay:                                                  # a "DataLoader" whose epochs are pre-made lists
        def __init__(self, per_epoch, which, n):
            self.per_epoch, self.which, self.e, self.n = per_epoch, which, 0, n
        def __len__(self):
            return self.n
        def __iter__(self):
            self.e += 1
            return iter(self.per_epoch[self.e - 1][self.which])

    def run(device_side, graph=False):
        model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="A549", task="active_E_vs_inactive_E", device=DEV,
                                        in_features_FFNN=F_in)
        torch.manual_seed(3)
        model.apply(ea.metrics.weight_reset)
        model.set_rng("philox", seed=5)
        opt = optim.Adam(model.parameters(), lr=1e-3)
        if device_side:
            train = data.device_loaders(xtr, str_, ytr, bs, DEV, balanced=True, random_state=123,
                                        feature_dtype=torch.float32, pack_sequence=pack)
            test = data.device_loaders(xte, ste, yte, 2 * bs, DEV, balanced=False, random_state=153,
                                       feature_dtype=torch.float32, pack_sequence=pack)
            assert ea.metrics.get_input_size(train["FFNN"]) == F_in
        else:
            s_tr = data.BalancedBatchSampler(ytr, bs, 123)
            s_te = data.ShuffledBatchSampler(len(yte), 2 * bs, 153)
            tr, te = host_lists(xtr, str_, ytr, s_tr), host_lists(xte, ste, yte, s_te)
            train = {"FFNN": Replay(tr, 0, len(s_tr)), "CNN": Replay(tr, 1, len(s_tr))}
            test = {"FFNN": Replay(te, 0, len(s_te)), "CNN": Replay(te, 1, len(s_te))}
        res = training.fit_multimodal(model, train, test, DEV, "A549", "active_E_vs_inactive_E", optimizer=opt,
                                      num_epochs=epochs, patience=10, verbose=False, precision="float32", graph=graph)
        return res, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    (ra, sa), (rb, sb), (rc, sc) = run(False), run(True), run(True, graph=True)
    assert ra[0] == rb[0] == rc[0] and ra[1] == rb[1] == rc[1]
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
        assert torch.equal(sa[k], sc[k]), (k, "graph-replayed steps on the loader's staging buffers")
