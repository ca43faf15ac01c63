"""Build-owned deterministic input generator (TEST INFRASTRUCTURE, not product).

Every golden vector, parity test and oracle run draws its inputs from here so
that the same bytes can be regenerated on any box without shipping tensors and
without depending on torch's (or numpy's) RNG stream stability.

The stream is splitmix64 keyed on a 64-bit FNV-1a hash of a tensor *name*:
    x_i = splitmix64(h(name) + i * 0x9E3779B97F4A7C15)
    uniform double = (x_i >> 11) * 2**-53            in [0, 1)

Layouts mirror what the reference's ``Dataset_Wrap`` emits
(BIOINF_tesi/data_pipe/dataprepare.py:398-412, data_pipe/utils.py:268-276):
    x_1    f64 [B, F]       epigenomic features, min-max scaled to [0, 1]
    x_2    f64 [B, 4, 256]  one-hot DNA window, channel order a,c,g,t
    target i64 [B, 1]
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def raw64(name: str, n: int) -> np.ndarray:
    """n splitmix64 outputs of the stream called `name` (uint64)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(_fnv1a(name)) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(name: str, shape, lo=0.0, hi=1.0) -> np.ndarray:
    """float64 uniforms in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (raw64(name, n) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return (lo + (hi - lo) * u).reshape(shape)


def integers(name: str, shape, n_values: int) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    return (raw64(name, n) % np.uint64(n_values)).astype(np.int64).reshape(shape)


def features(name: str, B: int, F: int) -> np.ndarray:
    return uniform(name, (B, F))


def onehot_sequence(name: str, B: int, L: int = 256) -> np.ndarray:
    base = integers(name, (B, L), 4)
    x = np.zeros((B, 4, L), dtype=np.float64)
    b = np.arange(B)[:, None]
    t = np.arange(L)[None, :]
    x[b, base, t] = 1.0
    return x


def labels(name: str, B: int, pos_rate: float) -> np.ndarray:
    return (uniform(name, (B,)) < pos_rate).astype(np.int64).reshape(B, 1)


def weight(name: str, shape, fan_in: int) -> np.ndarray:
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) -- same scale as nn.Linear/Conv1d default init."""
    k = 1.0 / np.sqrt(float(fan_in))
    return uniform(name, shape, -k, k)


def checksum(a: np.ndarray) -> dict:
    """Order-sensitive fp64 fingerprints used where a tensor is too large to commit."""
    a = np.asarray(a, dtype=np.float64).ravel()
    w = np.cos(np.arange(a.size, dtype=np.float64) * 0.61803398875)
    return {"sum": float(a.sum()), "abs": float(np.abs(a).sum()), "dot": float((a * w).sum()),
            "n": int(a.size)}
