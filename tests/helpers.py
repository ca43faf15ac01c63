"""Shared test helpers: golden-fixture access and regeneration of the inputs the fixtures were made from."""
import json
import os

import numpy as np

from oracle import datagen as dg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.arrays = {k: z[k] for k in z.files if k != "__meta__"}
        self.meta = json.loads(bytes(z["__meta__"]).decode())

    def __getitem__(self, k):
        return self.arrays[k]


def unpack_idx(packed, B, c):
    return np.unpackbits(packed)[: B * c].reshape(B, c).astype(np.int64)


def g1_inputs(case):
    """Inputs of a G1 case, regenerated exactly as tests/golden/make_golden.py did."""
    name, B, d0, d1, c = case["case"], case["B"], case["d0"], case["d1"], case["c"]
    X = [dg.uniform(name + "/x0", (B, d0)), dg.uniform(name + "/x1", (B, d1))]
    W = [dg.weight(name + "/w0", (c, d0), d0), dg.weight(name + "/w1", (c, d1), d1)]
    b = [dg.weight(name + "/b0", (c,), d0), dg.weight(name + "/b1", (c,), d1)]
    avail = None
    if case["avail"] == "onehot":
        avail = np.eye(2, dtype=np.float32)[dg.integers(name + "/avail_t", (B,), 2)]
    elif case["avail"] == "mixed":
        avail = np.array([[1, 1], [1, 0], [0, 1]], dtype=np.float32)[dg.integers(name + "/avail_t", (B,), 3)]
    p = None if case["p"] == "none" else dg.uniform(name + "/p", (B, 2), 0.05, 1.0).astype(np.float32)
    return X, W, b, avail, p


def g12_inputs(case):
    """Inputs of a G12 (bypass_docking) case, regenerated exactly as tests/golden/make_golden.py did."""
    name, B, c = case["case"], case["B"], case["c"]
    X = [dg.uniform(name + "/x0", (B, c), -1, 1), dg.uniform(name + "/x1", (B, c), -1, 1)]
    dout = dg.uniform(name + "/dout", (B, c), -1, 1)
    avail = None
    if case["avail"] == "mixed":
        avail = np.array([[1, 1], [1, 0], [0, 1]], dtype=np.float32)[dg.integers(name + "/avail_t", (B,), 3)]
    p = None if case["p"] == "none" else dg.uniform(name + "/p", (B, 2), 0.05, 1.0).astype(np.float32)
    return X, dout, avail, p


def g13_inputs(case):
    """Inputs of a G13 (M != 2 modalities) case, regenerated exactly as tests/golden/make_golden.py did."""
    name, B, ds, c, bypass = case["case"], case["B"], case["ds"], case["c"], case["bypass"]
    M = len(ds)
    X = [dg.uniform(f"{name}/x{m}", (B, d), -1, 1) for m, d in enumerate(ds)]
    W = [] if bypass else [dg.weight(f"{name}/w{m}", (c, d), d) for m, d in enumerate(ds)]
    b = [] if bypass else [dg.weight(f"{name}/b{m}", (c,), d) for m, d in enumerate(ds)]
    dout = dg.uniform(name + "/dout", (B, c), -1, 1)
    avail = None
    if case["avail"] == "mixed":
        a = dg.integers(name + "/avail", (B, M), 2).astype(np.float32)
        a[np.arange(B), dg.integers(name + "/avail_keep", (B,), M)] = 1.0
        avail = a
    p = None if case["p"] == "none" else dg.uniform(name + "/p", (B, M), 0.05, 1.0).astype(np.float32)
    return X, W, b, dout, avail, p


def model_fill(tag):
    """Parameter filler used for G2/G3/G9 models (same rule as make_golden.build_ref_model)."""
    def fill(key, shape):
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else max(int(shape[0]), 1)
        if key.endswith(".bias") or (len(shape) == 1):
            if ".CNN_model." in key and key.split(".")[2] != "0" and int(key.split(".")[2]) % 5 == 1:
                return dg.uniform(f"{tag}/{key}", shape, 0.5, 1.5) if key.endswith("weight") \
                    else dg.uniform(f"{tag}/{key}", shape, -0.2, 0.2)
            return dg.weight(f"{tag}/{key}", shape, 16)
        return dg.weight(f"{tag}/{key}", shape, fan)
    return fill


def model_batch(tag, B, F_in, pos_rate=0.1):
    return (dg.features(tag + "/x1", B, F_in), dg.onehot_sequence(tag + "/x2", B), dg.labels(tag + "/y", B, pos_rate))


class stock_prenets:
    """Context manager: run a model's two pre-networks on the STOCK torch operators their nn.Sequential members hold (the
    reference's own forward, FFNN_pre.py:47-49 / CNN_pre.py:72-76) instead of the HIP kernels -- the comparison side of the
    pre-network tests.  The fusion layer and the post stack keep running the HIP path."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        f, c = self.model.FFNN, self.model.CNN
        f.forward = lambda x, rng=None: f.model(x)
        c.forward = lambda x, rng=None: (lambda y: y.reshape(y.size(0), -1))(c.CNN_model(x))
        f.prelaunch = lambda x, rng=None: None
        return self.model

    def __exit__(self, *exc):
        for m in (self.model.FFNN, self.model.CNN):
            m.__dict__.pop("forward", None)
        self.model.FFNN.__dict__.pop("prelaunch", None)
        return False
