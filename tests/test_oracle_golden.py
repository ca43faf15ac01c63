"""CPU: the oracle (oracle/) against every golden fixture generated from the imported reference.
These pin the checker itself; they run without a GPU and without /root/reference."""
import numpy as np
import pytest
import torch

from helpers import Golden, g1_inputs, g12_inputs, g13_inputs, model_batch, model_fill, unpack_idx
from oracle import datagen as dg
from oracle import embrace_oracle as orc
from oracle import ref_step
from oracle.configs import CONFIGS


def test_datagen_is_stable():
    # first values of a named stream are part of the fixture contract
    assert dg.raw64("contract", 3).tolist() == [14664643891561012477, 13974651271544140241, 8120575280460097797]
    u = dg.uniform("contract", (4,))
    assert np.all((u >= 0) & (u < 1)) and len(set(u.tolist())) == 4
    x = dg.onehot_sequence("contract/x2", 3)
    assert x.shape == (3, 4, 256) and np.all(x.sum(1) == 1)


def test_g1_oracle_matches_reference_outputs():
    g = Golden("G1_embracenet_forward")
    for case in g.meta["cases"]:
        X, W, b, avail, p = g1_inputs(case)
        B, c = case["B"], case["c"]
        torch.manual_seed(case["seed"])
        u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
        cdf = orc.selection_cdf(np.ones((B, 2), np.float32) if p is None else p, avail)
        idx = orc.embrace_indices(cdf, u)
        assert np.array_equal(idx, unpack_idx(g[case["key"] + "_idx"], B, c)), case["tag"]
        npdt = np.float64 if case["dtype"] == "f64" else np.float32
        E, _ = orc.embrace_forward([x.astype(npdt) for x in X], [w.astype(npdt) for w in W], [v.astype(npdt) for v in b], idx)
        ref = g[case["key"] + "_out"].astype(np.float64)
        tol = 1e-12 if (case["dtype"] == "f64" and ref.dtype == np.float64 and B * c <= 4096) else 2e-5
        assert np.abs(E - ref).max() < tol, case["tag"]
        if case["dtype"] == "f64":
            chk = dg.checksum(E)
            assert abs(chk["sum"] - case["out_chk"]["sum"]) < 1e-9 * max(1.0, case["out_chk"]["abs"])


def _chk_equal(a, want):
    got = dg.checksum(a)
    return all(abs(got[k] - want[k]) <= 1e-12 * max(1.0, want["abs"]) for k in ("sum", "abs", "dot")) and got["n"] == want["n"]


def test_g12_oracle_matches_reference_bypass_docking():
    g = Golden("G12_bypass_docking")
    assert len(g.meta["cases"]) == 24
    for case in g.meta["cases"]:
        X, dout, avail, p = g12_inputs(case)
        B, c = case["B"], case["c"]
        torch.manual_seed(case["seed"])
        u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
        idx = orc.embrace_indices(orc.selection_cdf(np.ones((B, 2), np.float32) if p is None else p, avail), u)
        assert np.array_equal(idx, unpack_idx(g[case["key"] + "_idx"], B, c)), case["tag"]
        assert int(idx.sum()) == case["idx_ones"]
        npdt = np.float64 if case["dtype"] == "f64" else np.float32
        E = orc.embrace_bypass_forward([x.astype(npdt).astype(np.float64) for x in X], idx)
        dX = orc.embrace_bypass_backward(dout.astype(npdt).astype(np.float64), idx)
        assert _chk_equal(E, case["out_chk"]) and _chk_equal(dX[0], case["dx0_chk"]) and _chk_equal(dX[1], case["dx1_chk"]), case["tag"]


def test_g13_oracle_matches_reference_with_m_modalities():
    g = Golden("G13_m_modalities")
    assert len(g.meta["cases"]) == 30 and {len(c["ds"]) for c in g.meta["cases"]} == {1, 3, 4, 8}
    for case in g.meta["cases"]:
        X, W, b, dout, avail, p = g13_inputs(case)
        B, c, M = case["B"], case["c"], len(case["ds"])
        torch.manual_seed(case["seed"])
        u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
        idx = orc.embrace_indices(orc.selection_cdf(np.ones((B, M), np.float32) if p is None else p, avail), u)
        assert np.array_equal(idx, g[case["key"] + "_idx"]), case["tag"]
        npdt = np.float64 if case["dtype"] == "f64" else np.float32
        r = lambda a: a.astype(npdt).astype(np.float64)
        ref = g[case["key"] + "_out"].astype(np.float64)
        if case["bypass"]:
            E = orc.embrace_bypass_forward([r(x) for x in X], idx)
            dX = orc.embrace_bypass_backward(r(dout), idx, M)
        else:
            E, Z = orc.embrace_forward([r(x) for x in X], [r(w) for w in W], [r(v) for v in b], idx)
            dX, dW, db = orc.embrace_backward(r(dout), [r(x) for x in X], [r(w) for w in W], Z, idx)
        tol = 1e-12 if (case["dtype"] == "f64" and g[case["key"] + "_out"].dtype == np.float64) else 2e-5
        assert np.abs(E - ref).max() < tol, case["tag"]
        ftol = 1e-11 if case["dtype"] == "f64" else 2e-4
        for m in range(M):
            assert abs(dg.checksum(dX[m])["dot"] - case["dx_chk"][m]["dot"]) < ftol * max(1.0, case["dx_chk"][m]["abs"]), case["tag"]
            if not case["bypass"]:
                assert abs(dg.checksum(dW[m])["dot"] - case["dw_chk"][m]["dot"]) < ftol * max(1.0, case["dw_chk"][m]["abs"])
                assert abs(dg.checksum(db[m])["dot"] - case["db_chk"][m]["dot"]) < ftol * max(1.0, case["db_chk"][m]["abs"])


def test_g4_rng_contract():
    g = Golden("G4_rng_contract")
    for case in g.meta["cases"]:
        seed, B, c, p0 = case["seed"], case["B"], case["c"], case["p0"]
        torch.manual_seed(seed)
        torch.rand(1); torch.rand([B])
        u = torch.rand(B * c, dtype=torch.float64)
        assert np.array_equal(u[:64].numpy().view(np.uint64), g[case["key"] + "_u_bits"])
        p = np.repeat(np.array([[p0, 1 - p0]], np.float32), B, 0)
        cdf = orc.selection_cdf(p)
        assert int(cdf[0, 0].view(np.uint32)) == case["cdf0_bits"]
        assert np.array_equal(orc.embrace_indices(cdf, u.view(B, c).numpy()), unpack_idx(g[case["key"] + "_idx"], B, c))


def test_uniform53_definition():
    raw = np.array([0, 1, (1 << 53) - 1, (1 << 53), 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    u = orc.uniform53(raw)
    assert u[0] == 0 and u[1] == 2.0 ** -53 and u[2] == 1 - 2.0 ** -53 and u[3] == 0 and u[4] == 1 - 2.0 ** -53


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    z = np.zeros((1, 4), np.uint32)
    assert orc.philox4x32(z, np.zeros((1, 2), np.uint32))[0].tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    assert orc.philox4x32(f, np.full((1, 2), 0xFFFFFFFF, np.uint32))[0].tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    pi = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32)
    assert orc.philox4x32(pi, np.array([[0xa4093822, 0x299f31d0]], np.uint32))[0].tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_g5_weighted_ce():
    g = Golden("G5_weighted_ce")
    for case in g.meta["cases"]:
        i, B, rate = case["i"], case["B"], case["rate"]
        y = dg.labels(f"g5/{i}/y", B, rate) if 0 < rate < 1 else np.full((B, 1), int(rate), dtype=np.int64)
        z = dg.uniform(f"g5/{i}/z", (B, 2), -3, 3)
        loss, dz = orc.weighted_ce(z, y)
        assert abs(loss - case["loss"]) < 1e-6 and np.abs(dz - g[f"c{i}_dz"]).max() < 1e-6
        w = orc.class_weights(y)
        assert abs(w[0] - case["w_neg"]) < 1e-15 and abs(w[1] - case["w_pos"]) < 1e-15


def test_g6_metrics_closed_forms(ea):
    g = Golden("G6_metrics")
    for case in g.meta["cases"]:
        c = (case["tp"], case["pp"], case["p"], case["n"])
        assert abs(orc.batch_ap(*c) - case["ap"]) < 1e-12
        assert np.abs(orc.macro_prf(*c) - np.array(case["prf"])).max() < 1e-12
        # the package's own host-side closed forms (metrics.py) against the same sklearn-derived answers
        assert abs(ea.metrics.ap_from_counts(*c) - case["ap"]) < 1e-12
        assert np.abs(ea.metrics.prf_from_counts(*c) - np.array(case["prf"])).max() < 1e-12
        y = (dg.uniform(f"g6/{case['i']}/y", (case["B"], 1)) < case["rate"]).astype(np.int64)
        z = dg.uniform(f"g6/{case['i']}/z", (case["B"], 2), -1, 1)
        z[:, 1] += case["bias"]
        assert ea.metrics.confusion_counts(torch.from_numpy(z), torch.from_numpy(y)) == c
        assert abs(ea.AUPRC(torch.from_numpy(z), torch.from_numpy(y)) - case["ap"]) < 1e-12


def test_g7_optimizer_oracle():
    g = Golden("G7_optimizer_steps")
    for case in g.meta["cases"]:
        name = case["name"]
        p = dg.uniform(f"g7/{name}/p", (257,), -1, 1)
        m = np.zeros_like(p); v = np.zeros_like(p)
        for step in range(1, 4):
            gr = dg.uniform(f"g7/{name}/g{step}", (257,), -1, 1)
            if name == "adam":
                p, m, v = orc.adam_step(p, gr, m, v, step, case["lr"], case["weight_decay"])
            else:
                p, m = orc.rmsprop_step(p, gr, m, case["lr"], case["weight_decay"])
        assert np.abs(p - g[name + "_p3"]).max() < 1e-14


def test_nadam_oracle_equals_torch_nadam():
    """oracle.nadam_step restates timm.optim.Nadam (not installable here: timm parity stays formally unpinned).
    torch.optim.NAdam with momentum_decay = timm's schedule_decay and coupled weight decay implements the same published
    update (Dozat 2016), so it serves as an independent cross-check of the restatement."""
    import torch
    lr, wd, sd = 3e-3, 2e-2, 4e-3
    p0 = dg.uniform("nadam/cpu/p", (129,), -1, 1)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)       # torch keeps its mu_product scalar in the DEFAULT dtype (timm: a Python float)
    try:
        tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = torch.optim.NAdam([tp], lr=lr, weight_decay=wd, momentum_decay=sd)
        p, m, v, ms = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), 1.0
        for t in range(1, 6):
            g_ = dg.uniform(f"nadam/cpu/g{t}", (129,), -1, 1)
            tp.grad = torch.from_numpy(g_.copy())
            opt.step()
            p, m, v, ms = orc.nadam_step(p, g_, m, v, t, ms, lr, wd, schedule_decay=sd)
            assert np.abs(p - tp.detach().numpy()).max() < 1e-15, t
    finally:
        torch.set_default_dtype(prev)


def test_g8_error_cases():
    with pytest.raises(RuntimeError, match="invalid multinomial distribution"):
        orc.selection_cdf(np.array([[0.0, 1.0], [0.0, 1.0]]), np.array([[1.0, 0.0], [0.0, 1.0]]))


def _oracle_model(cfg, tag):
    hp, F_in = CONFIGS[cfg]
    m = ref_step.OracleEmbraceNetMultimodal(hp, F_in)
    m.set_tensors(model_fill(tag))
    return m, hp, F_in


def test_g2_full_model_oracle():
    g = Golden("G2_model_eval_logits")
    for case in g.meta["cases"]:
        m, hp, F_in = _oracle_model(case["cfg"], case["tag"])
        x1, x2, _ = model_batch(f"{case['tag']}/B{case['B']}", case["B"], F_in)
        m.eval()
        torch.manual_seed(case["seed"])
        out = m([torch.from_numpy(x1), torch.from_numpy(x2)])
        assert np.abs(out.detach().numpy() - g[case["key"] + "_logits"]).max() < 1e-12
        assert np.array_equal(m.last["idx"].numpy(), unpack_idx(g[case["key"] + "_idx"], case["B"], m.c))


def test_g10_inference_twin_oracle():
    """Row f3: softmax over the oracle's eval logits == the reference's EmbraceNetMultimodal_NoTrain, per region and batched."""
    g = Golden("G10_inference_twin")
    for case in g.meta["cases"]:
        m, hp, F_in = _oracle_model(case["cfg"], case["tag"])
        N = case["N"]
        x1, x2, _ = model_batch(f"{case['tag']}/N{N}", N, F_in)
        x1, x2 = torch.from_numpy(x1), torch.from_numpy(x2)
        m.eval()
        with torch.no_grad():
            # the reference loads the fp64 state dict into a freshly built fp32 module and only then calls .double()
            # (visual.py:278-279): the weights it predicts with are fp32-rounded
            for name in m.names:
                t = m.tensor(name)
                if t.is_floating_point():
                    t.copy_(t.float().double())
            torch.manual_seed(case["seed"])
            per = torch.stack([torch.softmax(m([x1[j:j + 1], x2[j:j + 1]]), dim=1).reshape(-1) for j in range(N)])
            torch.manual_seed(case["seed"] + 1)
            batched = torch.softmax(m([x1, x2]), dim=1).reshape(-1)
        assert np.abs(per.numpy() - g[case["key"] + "_per_sample"]).max() < 1e-12
        assert np.abs(batched.numpy() - g[case["key"] + "_batched"]).max() < 1e-12


def test_g3_train_step_oracle():
    g = Golden("G3_model_train_step")
    for case in g.meta["cases"]:
        m, hp, F_in = _oracle_model(case["cfg"], case["tag"])
        x1, x2, y = model_batch(f"{case['tag']}/B{case['B']}", case["B"], F_in)
        m.train()
        torch.manual_seed(case["seed"])
        out = m([torch.from_numpy(x1), torch.from_numpy(x2)], is_training=True)
        assert (m.last["t"] is not None) == case["dropped"] and abs(float(m.last["r"]) - case["r"]) < 1e-12
        loss = ref_step.batch_loss(out, torch.from_numpy(y))
        assert abs(loss.item() - case["loss"]) < 1e-7
        loss.backward()
        for name, chk in case["grads"].items():
            got = dg.checksum(m.tensor(name).grad.numpy())
            # conv biases in front of BatchNorm have a mathematically zero gradient (pure rounding noise): floor
            assert abs(got["sum"] - chk["sum"]) < max(1e-9 * chk["abs"], 1e-12), name


def test_g9_fit_trajectory_oracle():
    g = Golden("G9_fit_trajectory")
    for case in g.meta["cases"]:
        m, hp, F_in = _oracle_model("small", case["tag"])
        t = torch.from_numpy
        tr = [model_batch(f"{case['tag']}/train{k}", case["B"], F_in, 0.3) for k in range(case["n_train"])]
        te = [model_batch(f"{case['tag']}/test{k}", 2 * case["B"], F_in, 0.3) for k in range(case["n_test"])]
        ctor = torch.optim.Adam if case["opt"] == "adam" else torch.optim.RMSprop
        opt = ctor(m.parameters(), lr=case["lr"], weight_decay=case["weight_decay"])
        torch.manual_seed(case["seed"])
        losses, tr_sc, te_sc = [], [], []
        for ep in range(case["epochs"]):
            m.train(); s = 0.0
            for a, b, y in tr:
                l, ap = ref_step.train_step(m, opt, t(a), t(b), t(y)); losses.append(l); s += ap
            tr_sc.append(s / len(tr))
            m.eval(); s = 0.0
            for a, b, y in te:
                l, ap, out = ref_step.eval_step(m, t(a), t(b), t(y)); s += ap
            te_sc.append(s / len(te))
        assert np.allclose(tr_sc, case["AUPRC_train"], atol=1e-12) and np.allclose(te_sc, case["AUPRC_test"], atol=1e-12)
        assert np.abs(np.array(losses) - g[case["opt"] + "_losses"]).max() < 1e-9
        assert np.abs(m.tensor("embracenet.docking_0.weight").detach().numpy() - g[case["opt"] + "_dock0_w"]).max() < 1e-10
