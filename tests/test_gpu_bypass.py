"""GPU parity tests of EmbraceNet(bypass_docking=True) (reference: EmbraceNetMultimodal.py:54-55, 63-88) -- `pytest -m gpu`.

The layer is a select, so everything is bit-exact: indices, outputs and input gradients against fixture G12 (generated
from the imported reference) in host-replay mode, against the numpy oracle in Philox mode, and through size-independent
properties at a large size.
"""
import numpy as np
import pytest
import torch

from helpers import Golden, g12_inputs, unpack_idx
from oracle import datagen as dg
from oracle import embrace_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
TD = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def host(t):
    return t.detach().double().cpu().numpy()


def _chk_equal(a, want):
    got = dg.checksum(a)
    return all(abs(got[k] - want[k]) <= 1e-12 * max(1.0, want["abs"]) for k in ("sum", "abs", "dot")) and got["n"] == want["n"]


@pytest.mark.parametrize("i", range(24))
def test_g12_bypass_docking_matches_reference(ea, i):
    g = Golden("G12_bypass_docking")
    case = g.meta["cases"][i]
    X, dout, avail, p = g12_inputs(case)
    B, c, T = case["B"], case["c"], TD[case["dtype"]]
    net = ea.EmbraceNet(DEV, [c, c], c, bypass_docking=True).set_rng("host")
    assert len(list(net.parameters())) == 0 and not hasattr(net, "docking_0")         # :29-31
    xs = [dev(x, T).requires_grad_(True) for x in X]
    torch.manual_seed(case["seed"])
    out = net(xs, availabilities=None if avail is None else dev(avail), selection_probabilities=None if p is None else dev(p))
    (out * dev(dout, T)).sum().backward()
    idx = unpack_idx(g[case["key"] + "_idx"], B, c)
    assert np.array_equal(net.modality_indices().cpu().numpy(), idx), "multinomial index tensor not bit-exact"
    Xc = [host(x) for x in xs]
    assert np.array_equal(host(out), orc.embrace_bypass_forward(Xc, idx))
    want = orc.embrace_bypass_backward(host(dev(dout, T)), idx)
    assert np.array_equal(host(xs[0].grad), want[0]) and np.array_equal(host(xs[1].grad), want[1])
    assert _chk_equal(host(out), case["out_chk"])                                     # the reference's own output
    assert _chk_equal(host(xs[0].grad), case["dx0_chk"]) and _chk_equal(host(xs[1].grad), case["dx1_chk"])


@pytest.mark.parametrize("dt", ["f64", "f32", "bf16"])
@pytest.mark.parametrize("B,c", [(96, 200), (33, 30), (5, 7), (256, 768)])
def test_bypass_philox_mode_vs_oracle(ea, B, c, dt):
    F = ea.functional
    T, seed, step = TD[dt], 0xABCDEF0123, 9
    x0 = dev(dg.uniform(f"byp/{B}/{c}/x0", (B, c), -1, 1), T)
    x1 = dev(dg.uniform(f"byp/{B}/{c}/x1", (B, c), -1, 1), T)
    dE = dev(dg.uniform(f"byp/{B}/{c}/de", (B, c), -1, 1), T)
    p = dev(dg.uniform(f"byp/{B}/{c}/p", (B, 2), 0.05, 1.0).astype(np.float32))
    u = orc.philox_select_uniform(seed, (step << 8) | 0, np.arange(B * c, dtype=np.uint64) + np.uint64(17 * c)).reshape(B, c)
    idx = orc.embrace_indices(orc.selection_cdf(p.cpu().numpy()), u)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    cdf0, _ = F.select_prep(p, None, B)
    for sel in (cdf0, F.SelectInline(p, None, False, status)):           # prepared thresholds / thresholds in the launch
        a, b = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
        E, code = F.embrace_bypass(a, b, sel, rng=F.RngState(seed, step, row0=17))
        assert np.array_equal((code & 1).cpu().numpy(), idx)
        assert torch.all((code & 2) != 0)                                # no ReLU on this path: every element is active
        assert np.array_equal(host(E), orc.embrace_bypass_forward([host(x0), host(x1)], idx))
        E.backward(dE)
        want = orc.embrace_bypass_backward(host(dE), idx)
        assert np.array_equal(host(a.grad), want[0]) and np.array_equal(host(b.grad), want[1])
    assert int(status.item()) == 0
    # only one input needs a gradient: the other output pointer is NULL
    a = x0.clone().requires_grad_(True)
    E, _ = F.embrace_bypass(a, x1, cdf0, rng=F.RngState(seed, step, row0=17))
    E.backward(dE)
    assert np.array_equal(host(a.grad), want[0])


def test_bypass_properties_at_full_size(ea):
    """B = 4096, c = 1024 (the largest embracement size of the search space, cfg5's global batch), bf16."""
    F = ea.functional
    B, c = 4096, 1024
    g = torch.Generator(device="cpu").manual_seed(3)
    x0 = torch.randn(B, c, generator=g).to(DEV, torch.bfloat16)
    x1 = torch.randn(B, c, generator=g).to(DEV, torch.bfloat16)
    dE = torch.randn(B, c, generator=g).to(DEV, torch.bfloat16)
    a, b = x0.clone().requires_grad_(True), x1.clone().requires_grad_(True)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    p = torch.tensor([[0.25, 0.75]], device=DEV)
    E, code = F.embrace_bypass(a, b, F.SelectInline(p, None, False, status), rng=F.RngState(77, 3))
    E2, code2 = F.embrace_bypass(x0, x1, F.SelectInline(p, None, False, status), rng=F.RngState(77, 3))
    assert torch.equal(E, E2) and torch.equal(code, code2)               # idempotent for a fixed (seed, step)
    s1 = (code & 1).bool()
    assert torch.equal(E, torch.where(s1, x1, x0))
    frac = s1.float().mean().item()
    assert abs(frac - 0.75) < 2e-3                                       # 4.2 M draws: 3 sigma = 6.3e-4
    E.backward(dE)
    assert torch.equal(a.grad + b.grad, dE) and torch.equal(b.grad, torch.where(s1, dE, torch.zeros_like(dE)))
    # linearity in the inputs for a fixed selection
    E3, _ = F.embrace_bypass((x0.float() * 2).to(torch.bfloat16), (x1.float() * 2).to(torch.bfloat16),
                             F.SelectInline(p, None, False, status), rng=F.RngState(77, 3))
    assert torch.equal(E3.float(), E.detach().float() * 2)


def test_bypass_shape_and_modality_checks(ea):
    net = ea.EmbraceNet(DEV, [16, 16], 16, bypass_docking=True)
    with pytest.raises(ValueError, match="embracement_size"):
        net([torch.zeros(4, 16, device=DEV), torch.zeros(4, 8, device=DEV)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net([torch.zeros(4, 16), torch.zeros(4, 16)])
    net.set_rng("host")
    with pytest.raises(RuntimeError, match="invalid multinomial distribution"):
        net([torch.zeros(2, 16, device=DEV), torch.zeros(2, 16, device=DEV)],
            availabilities=torch.tensor([[1.0, 0.0], [0.0, 1.0]]), selection_probabilities=torch.tensor([[0.0, 1.0], [0.0, 1.0]]))


# ------------------------------------------------------------------------------------------ M != 2 modalities (G13)
TOLM = {"f64": 1e-11, "f32": 1e-5}


@pytest.mark.parametrize("i", range(30))
def test_g13_m_modalities_match_reference(ea, i):
    """len(input_list) in {1, 3, 4, 8} (EmbraceNetMultimodal.py:46-48), docking layers or bypass: indices bit-exact in
    host-replay mode, outputs and every gradient against the oracle and the reference's fingerprints."""
    from helpers import g13_inputs
    g = Golden("G13_m_modalities")
    case = g.meta["cases"][i]
    X, W, b, dout, avail, p = g13_inputs(case)
    B, c, M, dt = case["B"], case["c"], len(case["ds"]), case["dtype"]
    T = TD[dt]
    net = ea.EmbraceNet(DEV, case["ds"], c, bypass_docking=case["bypass"]).to(DEV).to(T).set_rng("host")
    if not case["bypass"]:
        with torch.no_grad():
            for m in range(M):
                getattr(net, f"docking_{m}").weight.copy_(dev(W[m], T))
                getattr(net, f"docking_{m}").bias.copy_(dev(b[m], T))
    xs = [dev(x, T).requires_grad_(True) for x in X]
    torch.manual_seed(case["seed"])
    out = net(xs, availabilities=None if avail is None else dev(avail), selection_probabilities=None if p is None else dev(p))
    (out * dev(dout, T)).sum().backward()
    idx = g[case["key"] + "_idx"].astype(np.int64)
    assert np.array_equal(net.modality_indices().cpu().numpy(), idx), "multinomial index tensor not bit-exact"
    assert np.bincount(idx.ravel(), minlength=M).tolist() == case["hist"]
    r = lambda a: host(dev(a, T))
    tol = TOLM[dt]
    if case["bypass"]:
        assert np.array_equal(host(out), orc.embrace_bypass_forward([r(x) for x in X], idx))
        dX = orc.embrace_bypass_backward(r(dout), idx, M)
        for m in range(M):
            assert np.array_equal(host(xs[m].grad), dX[m])
    else:
        E, Z = orc.embrace_forward([r(x) for x in X], [r(w) for w in W], [r(v) for v in b], idx)
        dX, dW, db = orc.embrace_backward(r(dout), [r(x) for x in X], [r(w) for w in W], Z, idx)
        assert np.abs(host(out) - E).max() < tol
        for m in range(M):
            lin = getattr(net, f"docking_{m}")
            scale = max(1.0, np.abs(dW[m]).max())
            assert np.abs(host(xs[m].grad) - dX[m]).max() < tol * 10
            assert np.abs(host(lin.weight.grad) - dW[m]).max() < tol * 10 * scale
            assert np.abs(host(lin.bias.grad) - db[m]).max() < tol * 10 * scale
    stored = g[case["key"] + "_out"]                            # the reference's output (large cases are stored as fp32)
    assert np.abs(host(out) - stored.astype(np.float64)).max() < (1e-11 if dt == "f64" and stored.dtype == np.float64 else 1e-5)
    if dt == "f64":
        assert _chk_equal(host(out), case["out_chk"]) or not case["bypass"]
        chk = dg.checksum(host(out))
        assert abs(chk["dot"] - case["out_chk"]["dot"]) < 1e-9 * max(1.0, case["out_chk"]["abs"])


@pytest.mark.parametrize("dt", ["f64", "f32", "bf16"])
@pytest.mark.parametrize("B,c,M", [(64, 256, 3), (33, 30, 5), (1024, 768, 4), (7, 8, 1)])
def test_m_modality_selection_philox_vs_oracle(ea, B, c, M, dt):
    F = ea.functional
    T, seed, step = TD[dt], 424242, 3
    xs = [dev(dg.uniform(f"selm/{B}/{c}/{M}/x{m}", (B, c), -1, 1), T) for m in range(M)]
    dE = dev(dg.uniform(f"selm/{B}/{c}/{M}/de", (B, c), -1, 1), T)
    p = dg.uniform(f"selm/{B}/{c}/{M}/p", (B, M), 0.05, 1.0).astype(np.float32)
    avail = dg.integers(f"selm/{B}/{c}/{M}/a", (B, M), 2).astype(np.float32)
    avail[:, 0] = 1.0
    want_cdf = orc.selection_cdf(p, avail)
    cdf, status = F.select_prep_m(dev(p), dev(avail), B, M)
    assert np.array_equal(cdf.cpu().numpy(), want_cdf) and int(status.item()) == 0      # fp32 op by op as ATen: bit-exact
    u = orc.philox_select_uniform(seed, (step << 8) | 0, np.arange(B * c, dtype=np.uint64) + np.uint64(5 * c)).reshape(B, c)
    idx = orc.embrace_indices(want_cdf, u)
    ins = [x.clone().requires_grad_(m != 1) for m, x in enumerate(xs)]                   # modality 1 needs no gradient
    E, code = F.embrace_select(ins, cdf, rng=F.RngState(seed, step, row0=5))
    assert np.array_equal(code.cpu().numpy().astype(np.int64), idx)
    assert np.array_equal(host(E), orc.embrace_bypass_forward([host(x) for x in xs], idx))
    E.backward(dE)
    want = orc.embrace_bypass_backward(host(dE), idx, M)
    for m in range(M):
        if m == 1:
            assert ins[m].grad is None
        else:
            assert np.array_equal(host(ins[m].grad), want[m])
    # M == 2 through the general entry equals the dedicated two-modality kernel
    if M >= 2:
        cdf2, _ = F.select_prep_m(dev(p[:, :2].copy()), None, B, 2)
        cdf0, _ = F.select_prep(dev(p[:, :2].copy()), None, B)
        assert torch.equal(cdf2[:, 0], cdf0)
        Ea, ca = F.embrace_select(xs[:2], cdf2, rng=F.RngState(seed, step))
        Eb, cb = F.embrace_bypass(xs[0], xs[1], cdf0, rng=F.RngState(seed, step))
        assert torch.equal(Ea, Eb) and torch.equal(ca, cb & 1)


def test_m_modalities_error_cases(ea):
    net = ea.EmbraceNet(DEV, [4, 4, 4], 8).to(DEV).set_rng("host")
    xs = [torch.zeros(2, 4, device=DEV)] * 3
    with pytest.raises(RuntimeError, match="invalid multinomial distribution"):
        net(xs, availabilities=torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 0.0]]))
    with pytest.raises(AssertionError):
        net(xs[:2])
    with pytest.raises(NotImplementedError):
        ea.EmbraceNet(DEV, [4] * 9, 8).to(DEV)([torch.zeros(2, 4, device=DEV)] * 9)
    big = ea.EmbraceNet(DEV, [4, 4, 4], 8).to(DEV)                                       # Philox mode, three modalities
    out = big([torch.rand(16, 4, device=DEV) for _ in range(3)])
    assert out.shape == (16, 8) and set(big.modality_indices().unique().tolist()) <= {0, 1, 2}
