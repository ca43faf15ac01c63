"""CPU oracle for the EmbraceNet fusion + classifier hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may.  It restates, in plain numpy (fp64 unless the reference itself uses
fp32), what the reference computes on its CPU path, function by function, each
citing the reference file:line it follows.  Reference root: /root/reference,
paths below are relative to ``BIOINF_tesi/models/``.

Pinning: every function here is checked against outputs of the *imported*
reference (tests/golden/make_golden.py, fixtures G1-G8 under tests/golden/).
``torch.multinomial``/``torch.rand`` themselves live in ATen (third-party,
torch 2.10.0, not in the reference tree); their published algorithm
(aten/src/ATen/native/cpu/MultinomialKernel.cpp, CPUGeneratorImpl mt19937,
uniform_real_distribution) is restated in `selection_cdf`, `embrace_indices`
and `uniform53`, and pinned by fixture G4.
"""
import numpy as np

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------- a3
def selection_cdf(p, avail=None):
    """EmbraceNetMultimodal.py:63-76 followed by ATen's multinomial cdf build.

    p      [B, M] or [M]  selection probabilities (any float dtype; cast to fp32 as :73 does)
    avail  [B, M] or None (ones, :64-65)
    returns cdf [B, M] float32, exactly the array ATen binary-searches.
    Raises RuntimeError when a row is not a valid distribution (p*a == 0 -> 0/0 = NaN),
    as torch.multinomial does.
    """
    p = np.asarray(p, dtype=F32)
    if p.ndim == 1:
        p = p[None, :]
    if avail is None:
        avail = np.ones_like(p)
    avail = np.asarray(avail, dtype=F32)
    if p.shape[0] == 1 and avail.shape[0] > 1:
        p = np.repeat(p, avail.shape[0], axis=0)           # :184 .repeat(B, 1)
    q = (p * avail).astype(F32)                             # :73
    s = np.zeros(q.shape[0], dtype=F32)
    for m in range(q.shape[1]):                             # torch.sum over M in fp32, :75
        s = (s + q[:, m]).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (q / s[:, None]).astype(F32)                    # :76
    if not np.all(np.isfinite(q)) or np.any(q < 0):
        raise RuntimeError("invalid multinomial distribution (encountering probability entry = infinity or NaN)")
    # ATen multinomial_with_replacement: running fp32 sum, then cum /= sum
    cum = np.zeros_like(q)
    run = np.zeros(q.shape[0], dtype=F32)
    for m in range(q.shape[1]):
        run = (run + q[:, m]).astype(F32)
        cum[:, m] = run
    if np.any(run <= 0):
        raise RuntimeError("invalid multinomial distribution (sum of probabilities <= 0)")
    cum = (cum / run[:, None]).astype(F32)
    return cum


# --------------------------------------------------------------------------- a5
def uniform53(raw64):
    """at::uniform_real_distribution<double>(0,1) on one mt19937 random64():
    (x & (2^53 - 1)) * 2^-53."""
    raw64 = np.asarray(raw64, dtype=np.uint64)
    return (raw64 & np.uint64((1 << 53) - 1)).astype(F64) * (2.0 ** -53)


def embrace_indices(cdf, u):
    """EmbraceNetMultimodal.py:84 -- torch.multinomial(p, c, replacement=True) on CPU.

    ATen draws one fp64 uniform per (row, sample), row-major, and binary-searches the fp32
    cdf for the first slot with cdf[slot] >= u (compare in fp64).  For a non-decreasing cdf
    that slot is the number of entries strictly below u.
    cdf [B, M] float32, u [B, c] float64  ->  idx [B, c] int64
    """
    cdf = np.asarray(cdf, dtype=F32).astype(F64)
    u = np.asarray(u, dtype=F64)
    idx = np.zeros(u.shape, dtype=np.int64)
    for m in range(cdf.shape[1] - 1):        # last bin is 1.0 >= any u in [0,1)
        idx += (cdf[:, m][:, None] < u)
    return idx


# ----------------------------------------------------------------------- a2,a4-a7
def embrace_forward(X, W, b, idx, dtype=F64):
    """EmbraceNetMultimodal.py:52-60 (docking + ReLU), :80-88 (stack, one-hot, mul, sum).

    X[m] [B, d_m], W[m] [c, d_m] (nn.Linear layout), b[m] [c], idx [B, c].
    returns E [B, c] and the list of pre-activations Z_m.
    The masked sum over modalities has exactly one non-zero term, so it is a select.
    """
    Z = [np.asarray(x, dtype).dot(np.asarray(w, dtype).T) + np.asarray(bb, dtype)
         for x, w, bb in zip(X, W, b)]
    D = [np.maximum(z, 0) for z in Z]
    E = np.zeros_like(D[0])
    for m, d in enumerate(D):
        E = np.where(idx == m, d, E)
    return E, Z


def embrace_backward(dE, X, W, Z, idx):
    """Autograd of the above (reference: loss.backward(), training_models_multimodal.py:156).
    returns lists dX, dW, db."""
    dX, dW, db = [], [], []
    for m, (x, w, z) in enumerate(zip(X, W, Z)):
        dD = dE * (idx == m) * (z > 0)
        dW.append(dD.T.dot(x))
        db.append(dD.sum(0))
        dX.append(dD.dot(w))
    return dX, dW, db


def embrace_bypass_forward(X, idx):
    """EmbraceNetMultimodal.py:54-55 (bypass_docking: docking_output_list = input_list), :80-88 (stack, one-hot, mul, sum).
    X[m] [B, c], idx [B, c]  ->  E [B, c] = X[idx[b, j]][b, j]."""
    E = np.zeros_like(np.asarray(X[0]))
    for m, x in enumerate(X):
        E = np.where(idx == m, x, E)
    return E


def embrace_bypass_backward(dE, idx, M=2):
    """Autograd of the above: dX_m = dE * one_hot(idx)[..., m]."""
    return [dE * (idx == m) for m in range(M)]


# --------------------------------------------------------------------------- a10
def linear_forward(x, w, b, relu):
    """nn.Linear (+ nn.ReLU) of the post stack, EmbraceNetMultimodal.py:143-151."""
    z = x.dot(w.T) + b
    return (np.maximum(z, 0) if relu else z), z


def linear_backward(dy, x, w, z, relu):
    if relu:
        dy = dy * (z > 0)
    return dy.dot(w), dy.T.dot(x), dy.sum(0)


# --------------------------------------------------------------------------- a11
def class_weights(target):
    """utils/utils.py:121-140 + the [w_neg, w_pos] reorder at training_models_multimodal.py:141.
    returns weight[2] = [pos/B, neg/B] (with the pos==0 / neg==0 special cases)."""
    t = np.asarray(target).reshape(-1)
    pos = int((t == 1).sum())
    neg = int((t == 0).sum())
    pos_inv = 1.0 / pos if pos != 0 else 0.0
    neg_inv = 1.0 / neg if neg != 0 else 0.0
    w_pos = pos_inv / (neg_inv + pos_inv)
    w_neg = neg_inv / (neg_inv + pos_inv)
    return np.array([w_neg, w_pos], dtype=F64)


def weighted_ce(logits, target, weight=None, dtype=F32):
    """nn.CrossEntropyLoss(weight)(output.float(), target.squeeze()),
    training_models_multimodal.py:141,151-154 (the fp32 branch is the one that runs).
    returns (loss, dlogits)."""
    z = np.asarray(logits, dtype=dtype)
    t = np.asarray(target).reshape(-1)
    w = class_weights(t).astype(dtype) if weight is None else np.asarray(weight, dtype)
    zmax = z.max(1, keepdims=True)
    lse = zmax[:, 0] + np.log(np.exp(z - zmax).sum(1))
    nll = lse - z[np.arange(len(t)), t]
    wy = w[t]
    den = wy.sum()
    loss = (wy * nll).sum() / den
    sm = np.exp(z - lse[:, None])
    oh = np.zeros_like(z)
    oh[np.arange(len(t)), t] = 1
    dz = (sm - oh) * (wy / den)[:, None]
    return loss, dz


# ----------------------------------------------------------------- metrics (G6)
def confusion_counts(logits, target):
    pred = np.argmax(np.asarray(logits), axis=1)
    t = np.asarray(target).reshape(-1)
    tp = int(((pred == 1) & (t == 1)).sum())
    pp = int((pred == 1).sum())
    p = int((t == 1).sum())
    return tp, pp, p, len(t)


def batch_ap(tp, pp, p, n):
    """utils/utils.py:80-86: sklearn average_precision_score on hard argmax predictions has a
    closed form in (TP, predicted-positive, positive, n)."""
    if p == 0:
        return 0.0
    if pp == 0 or pp == n:
        return p / n
    r = tp / p
    return r * (tp / pp) + (1 - r) * (p / n)


def macro_prf(tp, pp, p, n):
    """utils/utils.py:89-94: precision_recall_fscore_support(average='macro', zero_division=0)[:3]
    for labels present in target or prediction."""
    fn = p - tp
    fp = pp - tp
    tn = n - tp - fn - fp
    per = []
    for (t_, f_p, f_n, present) in ((tp, fp, fn, (p > 0) or (pp > 0)),
                                    (tn, fn, fp, (n - p > 0) or (n - pp > 0))):
        if not present:
            continue
        prec = t_ / (t_ + f_p) if (t_ + f_p) > 0 else 0.0
        rec = t_ / (t_ + f_n) if (t_ + f_n) > 0 else 0.0
        f1 = 2 * prec * rec / (prec + rec) if (prec + rec) > 0 else 0.0
        per.append((prec, rec, f1))
    a = np.array(per)
    return a.mean(0)


# ------------------------------------------------------------- counter RNG (perf)
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32(counter, key, rounds=10):
    """Philox4x32-10 (Salmon et al., SC'11).  counter: uint32 [..., 4], key: uint32 [..., 2]."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0 = _PHILOX_M0 * c[0]
        p1 = _PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(_PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(_PHILOX_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def philox_words(seed, stream, index):
    """The package's RNG contract (csrc/philox.h): key = seed (lo,hi), counter =
    (index lo, index hi, stream lo, stream hi).  index: uint64 array."""
    index = np.asarray(index, dtype=np.uint64)
    ctr = np.stack([(index & np.uint64(0xFFFFFFFF)), (index >> np.uint64(32)),
                    np.full(index.shape, stream & 0xFFFFFFFF, dtype=np.uint64),
                    np.full(index.shape, (stream >> 32) & 0xFFFFFFFF, dtype=np.uint64)],
                   axis=-1).astype(np.uint32)
    key = np.stack([np.full(index.shape, seed & 0xFFFFFFFF, dtype=np.uint64),
                    np.full(index.shape, (seed >> 32) & 0xFFFFFFFF, dtype=np.uint64)],
                   axis=-1).astype(np.uint32)
    return philox4x32(ctr, key)


def philox_uniform53(seed, stream, index):
    w = philox_words(seed, stream, index).astype(np.uint64)
    raw = w[..., 0] | (w[..., 1] << np.uint64(32))
    return uniform53(raw)


def philox_select_uniform(seed, stream, element):
    """Perf-mode selection uniform of element e = global_row * c + j (RNG kind 0, include/embrace_hip.h): word (e & 3) of
    Philox(counter = e >> 2) times 2^-32 -- one Philox call serves four consecutive elements.  Exact in fp64, so
    embrace_indices(cdf, u) on it is the kernels' integer compare floor(cdf0 * 2^32) < word."""
    e = np.asarray(element, dtype=np.uint64)
    w = philox_words(seed, stream, e >> np.uint64(2))
    k = (e & np.uint64(3)).astype(np.int64)
    word = np.take_along_axis(w, k[..., None], axis=-1)[..., 0]
    return word.astype(F64) * (2.0 ** -32)


def philox_uniform24(seed, stream, index):
    """fp32 uniform with the same construction as at::uniform_real_distribution<float>:
    (x & (2^24 - 1)) * 2^-24."""
    w = philox_words(seed, stream, index)
    return ((w[..., 0] & np.uint32((1 << 24) - 1)).astype(F32) * F32(2.0 ** -24)).astype(F32)


# ---------------------------------------------------------------- optimizers (a13)
def adam_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (coupled L2), training_models_multimodal.py:325."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
    return p, m, v


def rmsprop_step(p, g, sq, lr, wd, alpha=0.99, eps=1e-8):
    """torch.optim.RMSprop defaults (momentum 0, not centered)."""
    g = g + wd * p
    sq = alpha * sq + (1 - alpha) * g * g
    p = p - lr * g / (np.sqrt(sq) + eps)
    return p, sq


def nadam_step(p, g, m, v, step, m_schedule, lr, wd, b1=0.9, b2=0.999, eps=1e-8,
               schedule_decay=4e-3):
    """timm.optim.Nadam (absent from this image; restated from its published algorithm,
    Dozat 2016 as implemented in timm<=0.4/0.5: warm momentum schedule
    mu_t = b1*(1 - 0.5*0.96^(t*schedule_decay))).  PARITY UNPINNED."""
    g = g + wd * p
    mu_t = b1 * (1.0 - 0.5 * (0.96 ** (step * schedule_decay)))
    mu_next = b1 * (1.0 - 0.5 * (0.96 ** ((step + 1) * schedule_decay)))
    ms_new = m_schedule * mu_t
    ms_next = ms_new * mu_next
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    v_hat = v / (1 - b2 ** step)
    denom = np.sqrt(v_hat) + eps
    p = p - lr * (1 - mu_t) / (1 - ms_new) * g / denom
    p = p - lr * mu_next / (1 - ms_next) * m / denom
    return p, m, v, ms_new
