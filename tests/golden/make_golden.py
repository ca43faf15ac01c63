#!/usr/bin/env python3
"""Generate the golden fixtures G1-G13 by importing the reference on CPU.

Runs ONLY in the build container (needs /root/reference).  The fixtures it writes under
tests/golden/ are data (inputs are regenerated from oracle/datagen.py by name; outputs are
stored), never reference source.  Re-run with:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import recipe = SURVEY.md 8c: the reference's package __init__ chain imports six third-party
modules that are absent here and unused by the hot path; inert stubs stand in for them.
While generating, every fixture is also cross-checked against oracle/ (the CPU restatement),
so a committed fixture implies "oracle == reference" on that case at generation time.
"""
import io
import json
import os
import sys
import tempfile
import types
import contextlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        m = _Stub(f"{self.__name__}.{n}")
        sys.modules[m.__name__] = m
        setattr(self, n, m)
        return m

    def __call__(self, *a, **k):
        return self


for _n in ["seaborn", "optuna", "optuna.integration", "optuna.samplers", "timm", "timm.optim",
           "botorch", "imblearn", "imblearn.over_sampling", "miceforest"]:
    sys.modules[_n] = _Stub(_n)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from BIOINF_tesi.models.EmbraceNetMultimodal import EmbraceNet, EmbraceNetMultimodal  # noqa: E402
from BIOINF_tesi.models.utils.training_models_multimodal import fit_multimodal  # noqa: E402
from BIOINF_tesi.models.utils import utils as ref_utils  # noqa: E402

from oracle import datagen as dg  # noqa: E402
from oracle import embrace_oracle as orc  # noqa: E402
from oracle import ref_step  # noqa: E402
from oracle.configs import CONFIGS, FixedTrial  # noqa: E402

torch.set_num_threads(8)


def packbits(idx):
    return np.packbits(np.asarray(idx, dtype=np.uint8).ravel())


def save(name, arrays, meta):
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in arrays.values())/1024:.1f} KiB raw)")


# =========================================================================== G1
def avail_variant(kind, name, B):
    if kind == "none":
        return None
    if kind == "onehot":
        t = dg.integers(name + "/avail_t", (B,), 2)
        return np.eye(2, dtype=np.float32)[t]
    if kind == "mixed":                       # rows: both / only-0 / only-1
        t = dg.integers(name + "/avail_t", (B,), 3)
        return np.array([[1, 1], [1, 0], [0, 1]], dtype=np.float32)[t]
    raise ValueError(kind)


def g1():
    arrays, meta = {}, {"cases": []}
    shapes = [(8, 4, 64, 32), (64, 16, 1856, 512), (64, 256, 7936, 1024), (37, 32, 1024, 768)]
    for (B, d0, d1, c) in shapes:
        for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            for av_kind, p_kind in (("none", "none"), ("onehot", "given"), ("mixed", "given"), ("none", "given")):
                if B == 64 and d1 == 7936 and (dt_name, av_kind) not in (("f64", "none"), ("f32", "mixed")):
                    continue
                case = f"g1/B{B}_d{d0}_{d1}_c{c}"
                tag = f"{case}/{dt_name}/{av_kind}/{p_kind}"
                seed = 1000 + len(meta["cases"])
                X = [dg.uniform(case + "/x0", (B, d0)), dg.uniform(case + "/x1", (B, d1))]
                W = [dg.weight(case + "/w0", (c, d0), d0), dg.weight(case + "/w1", (c, d1), d1)]
                b = [dg.weight(case + "/b0", (c,), d0), dg.weight(case + "/b1", (c,), d1)]
                avail = avail_variant(av_kind, case, B)
                p = None if p_kind == "none" else dg.uniform(case + "/p", (B, 2), 0.05, 1.0).astype(np.float32)
                net = EmbraceNet("cpu", [d0, d1], c).to(dt)
                with torch.no_grad():
                    for m in range(2):
                        getattr(net, f"docking_{m}").weight.copy_(torch.from_numpy(W[m]).to(dt))
                        getattr(net, f"docking_{m}").bias.copy_(torch.from_numpy(b[m]).to(dt))
                torch.manual_seed(seed)
                out = net([torch.from_numpy(x).to(dt) for x in X],
                          availabilities=None if avail is None else torch.from_numpy(avail),
                          selection_probabilities=None if p is None else torch.from_numpy(p))
                # replay of the generator: the B*c fp64 uniforms torch.multinomial consumed
                torch.manual_seed(seed)
                u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
                # oracle cross-check
                cdf = orc.selection_cdf(np.ones((B, 2), np.float32) if p is None else p, avail)
                idx = orc.embrace_indices(cdf, u)
                npdt = np.float64 if dt_name == "f64" else np.float32
                E, _ = orc.embrace_forward([x.astype(npdt) for x in X], [w.astype(npdt) for w in W],
                                           [bb.astype(npdt) for bb in b], idx, dtype=np.float64)
                ref = out.detach().double().numpy()
                tol = 1e-12 if dt_name == "f64" else 2e-5
                err = np.abs(E - ref).max()
                assert err < tol, (tag, err)
                # the reference never returns idx; recover it from which docking output was selected
                D = [np.maximum(X[m].astype(npdt).astype(np.float64) @ W[m].astype(npdt).astype(np.float64).T
                                + b[m].astype(npdt), 0) for m in range(2)]
                amb = np.isclose(D[0], D[1], atol=1e-9)
                ref_idx = (np.abs(ref - D[1]) < np.abs(ref - D[0])).astype(np.int64)
                assert np.all((ref_idx == idx) | amb), tag
                key = tag.replace("/", "_")
                arrays[key + "_idx"] = packbits(idx)
                arrays[key + "_out"] = ref.astype(np.float32) if B * c > 4096 else ref
                meta["cases"].append(dict(tag=tag, key=key, case=case, B=B, d0=d0, d1=d1, c=c, dtype=dt_name,
                                          avail=av_kind, p=p_kind, seed=seed, out_chk=dg.checksum(ref),
                                          idx_ones=int(idx.sum()), oracle_err=float(err)))
                print("G1", tag, "err", err)
    save("G1_embracenet_forward", arrays, meta)


# ======================================================================= helpers
def build_ref_model(cfg_name, tag):
    hp, F_in = CONFIGS[cfg_name]
    trial = FixedTrial(hp)
    model = EmbraceNetMultimodal(trial, cell_line="A549", task="active_E_vs_inactive_E", device="cpu",
                                 in_features_FFNN=F_in)
    model = model.double()
    sd = model.state_dict()

    def fill(key, shape):
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else max(int(shape[0]), 1)
        if key.endswith(".bias") or (len(shape) == 1):
            if ".CNN_model." in key and key.split(".")[2] != "0" and int(key.split(".")[2]) % 5 == 1:
                # BatchNorm affine: gamma ~ U(0.5,1.5), beta ~ U(-0.2,0.2)
                return dg.uniform(f"{tag}/{key}", shape, 0.5, 1.5) if key.endswith("weight") \
                    else dg.uniform(f"{tag}/{key}", shape, -0.2, 0.2)
            return dg.weight(f"{tag}/{key}", shape, 16)
        return dg.weight(f"{tag}/{key}", shape, fan)

    with torch.no_grad():
        for key, t in sd.items():
            if "running_" in key or "num_batches" in key:
                continue
            t.copy_(torch.from_numpy(fill(key, tuple(t.shape))))
    oracle = ref_step.OracleEmbraceNetMultimodal(hp, F_in)
    oracle.set_tensors(fill)
    for key in oracle.names:
        if "running_" not in key:
            assert torch.equal(oracle.tensor(key), sd[key]), key
    return model, oracle, trial, hp, F_in


def batch(tag, B, F_in, pos_rate=0.1):
    x1 = dg.features(tag + "/x1", B, F_in)
    x2 = dg.onehot_sequence(tag + "/x2", B)
    y = dg.labels(tag + "/y", B, pos_rate)
    return torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(y)


# =========================================================================== G2
def g2():
    arrays, meta = {}, {"cases": []}
    for cfg_name, B, seed in (("cfg1", 64, 2001), ("post2", 64, 2002), ("small", 100, 2003), ("cfg1", 37, 2004)):
        tag = f"g2/{cfg_name}"
        model, oracle, trial, hp, F_in = build_ref_model(cfg_name, tag)
        x1, x2, y = batch(f"{tag}/B{B}", B, F_in)
        model.eval(); oracle.eval()
        torch.manual_seed(seed)
        out = model([x1, x2])
        torch.manual_seed(seed)
        out_o = oracle([x1, x2])
        err = (out - out_o).abs().max().item()
        assert err < 1e-12, (tag, err)
        key = f"{cfg_name}_B{B}"
        arrays[key + "_logits"] = out.detach().numpy()
        arrays[key + "_idx"] = packbits(oracle.last["idx"].numpy())
        arrays[key + "_E"] = oracle.last["E"].detach().numpy().astype(np.float32)
        meta["cases"].append(dict(key=key, cfg=cfg_name, tag=tag, B=B, seed=seed, oracle_err=err,
                                  trial_calls=trial.calls, n_params=sum(p.numel() for p in model.parameters()),
                                  E_chk=dg.checksum(oracle.last["E"].detach().numpy())))
        print("G2", key, "err", err, "params", meta["cases"][-1]["n_params"])
    save("G2_model_eval_logits", arrays, meta)


# =========================================================================== G3
def g3():
    arrays, meta = {}, {"cases": []}
    want = {("cfg1", True): None, ("cfg1", False): None, ("post2", True): None, ("small", False): None}
    for cfg_name in ("cfg1", "post2", "small"):
        tag = f"g3/{cfg_name}"
        B = 64
        for seed in range(3000, 3040):
            model, oracle, trial, hp, F_in = build_ref_model(cfg_name, tag)
            x1, x2, y = batch(f"{tag}/B{B}", B, F_in)
            model.train(); oracle.train()
            cap = {}

            def hook(mod, args, kwargs):
                cap["in"] = [t for t in args[0]]
                for t in cap["in"]:
                    t.retain_grad()
            h = model.embracenet.register_forward_pre_hook(hook, with_kwargs=True)
            torch.manual_seed(seed)
            out = model([x1, x2], is_training=True)
            h.remove()
            torch.manual_seed(seed)
            out_o = oracle([x1, x2], is_training=True)
            dropped = oracle.last["t"] is not None
            if (cfg_name, dropped) not in want or want[(cfg_name, dropped)] is not None:
                continue
            want[(cfg_name, dropped)] = seed
            # reference loss: the except-branch of training_models_multimodal.py:151-154
            w_pos, w_neg = ref_utils.get_loss_weights_from_labels(y)
            crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([w_neg, w_pos]))
            loss = crit.float()(out.float(), y.squeeze())
            loss.backward()
            oracle.last["h0"].retain_grad(); oracle.last["h1"].retain_grad()
            loss_o = ref_step.batch_loss(out_o, y)
            loss_o.backward()
            err = (out - out_o).abs().max().item()
            assert err < 1e-12 and abs(loss.item() - loss_o.item()) < 1e-7, (tag, err)
            key = f"{cfg_name}_{'drop' if dropped else 'keep'}"
            grads = {}
            sd_params = dict(model.named_parameters())
            for k in oracle.names:
                if "running_" in k:
                    continue
                g_ref = sd_params[k].grad.numpy()
                g_o = oracle.tensor(k).grad.numpy()
                assert np.abs(g_ref - g_o).max() < 1e-10 * max(1.0, np.abs(g_ref).max()), k
                grads[k] = dg.checksum(g_ref)
            dX = [t.grad.numpy() for t in cap["in"]]
            assert np.abs(dX[0] - oracle.last["h0"].grad.numpy()).max() < 1e-12
            assert np.abs(dX[1] - oracle.last["h1"].grad.numpy()).max() < 1e-12
            arrays[key + "_logits"] = out.detach().numpy()
            arrays[key + "_idx"] = packbits(oracle.last["idx"].numpy())
            arrays[key + "_dX0"] = dX[0]
            arrays[key + "_dX1"] = dX[1].astype(np.float32)
            arrays[key + "_g_dock0_w"] = sd_params["embracenet.docking_0.weight"].grad.numpy()
            arrays[key + "_g_dock0_b"] = sd_params["embracenet.docking_0.bias"].grad.numpy()
            arrays[key + "_g_dock1_b"] = sd_params["embracenet.docking_1.bias"].grad.numpy()
            if dropped:
                arrays[key + "_t"] = oracle.last["t"].numpy().astype(np.uint8)
            bn_keys = [k for k in model.state_dict() if "running_" in k]
            for k in bn_keys:
                arrays[key + "_bn_" + k.replace(".", "_")] = model.state_dict()[k].numpy()
            meta["cases"].append(dict(key=key, cfg=cfg_name, tag=tag, B=B, seed=seed, dropped=bool(dropped),
                                      r=float(oracle.last["r"]), loss=float(loss.item()), grads=grads,
                                      dX1_chk=dg.checksum(dX[1]), bn_keys=bn_keys, oracle_err=err))
            print("G3", key, "seed", seed, "r", float(oracle.last["r"]), "loss", loss.item())
        # all wanted branches of this cfg found?
    missing = [k for k, v in want.items() if v is None]
    assert not missing, missing
    save("G3_model_train_step", arrays, meta)


# =========================================================================== G4
def g4():
    arrays, meta = {}, {"cases": []}
    for i, (seed, B, c, p0) in enumerate([(0, 4, 16, 0.5), (1, 64, 512, 0.5784523087676721), (123, 32, 1024, 0.07),
                                          (7, 5, 24, 0.999), (11, 3, 8, 1e-4)]):
        p = torch.tensor([[p0, 1 - p0]], dtype=torch.float32).repeat(B, 1)
        torch.manual_seed(seed)
        r1 = torch.rand(1)                               # the fp32 draws of EmbraceNetMultimodal.py:179,181
        rB = torch.rand([B])
        st = torch.get_rng_state()
        idx = torch.multinomial(p, num_samples=c, replacement=True)
        end_state = torch.get_rng_state()
        torch.set_rng_state(st)
        u = torch.rand(B * c, dtype=torch.float64)
        assert torch.equal(torch.get_rng_state(), end_state), "generator end-state differs"
        cdf = orc.selection_cdf(p.numpy())
        idx_o = orc.embrace_indices(cdf, u.view(B, c).numpy())
        assert np.array_equal(idx_o, idx.numpy()), (seed, B, c)
        key = f"s{seed}_B{B}_c{c}"
        arrays[key + "_idx"] = packbits(idx.numpy())
        arrays[key + "_u_bits"] = u[:64].numpy().view(np.uint64)
        arrays[key + "_r1_bits"] = r1.numpy().view(np.uint32)
        arrays[key + "_rB_bits"] = rB.numpy().view(np.uint32)
        meta["cases"].append(dict(key=key, seed=seed, B=B, c=c, p0=p0, idx_ones=int(idx.sum()),
                                  cdf0_bits=int(cdf[0, 0].view(np.uint32))))
        print("G4", key, "ones", int(idx.sum()))
    save("G4_rng_contract", arrays, meta)


# =========================================================================== G5
def g5():
    arrays, meta = {}, {"cases": []}
    for i, (B, rate) in enumerate([(64, 0.1), (64, 0.0), (64, 1.0), (100, 0.5), (7, 0.3), (1, 1.0), (200, 0.02)]):
        y = torch.from_numpy(dg.labels(f"g5/{i}/y", B, rate)) if 0 < rate < 1 else \
            torch.full((B, 1), int(rate), dtype=torch.int64)
        z = torch.from_numpy(dg.uniform(f"g5/{i}/z", (B, 2), -3, 3))
        z.requires_grad_(True)
        w_pos, w_neg = ref_utils.get_loss_weights_from_labels(y)
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([w_neg, w_pos]))
        zf = z.float()
        zf.retain_grad()
        loss = crit.float()(zf, y.squeeze(1) if B > 1 else y.reshape(1))
        loss.backward()
        lo, dz = orc.weighted_ce(z.detach().numpy(), y.numpy())
        assert abs(lo - loss.item()) < 1e-6 and np.abs(dz - zf.grad.numpy()).max() < 1e-6
        wo = orc.class_weights(y.numpy())
        assert abs(wo[0] - w_neg) < 1e-15 and abs(wo[1] - w_pos) < 1e-15
        arrays[f"c{i}_dz"] = zf.grad.numpy()
        meta["cases"].append(dict(i=i, B=B, rate=rate, loss=float(loss.item()), w_neg=float(w_neg),
                                  w_pos=float(w_pos), pos=int(y.sum())))
        print("G5", i, B, rate, loss.item(), w_neg, w_pos)
    save("G5_weighted_ce", arrays, meta)


# =========================================================================== G6
def g6():
    meta = {"cases": []}
    for i in range(60):
        B = [1, 2, 5, 37, 64, 100, 200][i % 7]
        rate = [0.0, 0.1, 0.5, 0.9, 1.0][i % 5]
        y = torch.from_numpy((dg.uniform(f"g6/{i}/y", (B, 1)) < rate).astype(np.int64))
        bias = [-4.0, -1.0, 0.0, 1.0, 4.0][(i // 5) % 5]
        z = torch.from_numpy(dg.uniform(f"g6/{i}/z", (B, 2), -1, 1))
        z[:, 1] += bias
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ap = float(ref_utils.AUPRC(z, y))
            prf = ref_utils.F1_precision_recall(z, y.reshape(-1)).tolist()
        tp, pp, p, n = orc.confusion_counts(z.numpy(), y.numpy())
        assert abs(orc.batch_ap(tp, pp, p, n) - ap) < 1e-12, (i, tp, pp, p, n, ap)
        assert np.abs(orc.macro_prf(tp, pp, p, n) - np.array(prf)).max() < 1e-12, (i, tp, pp, p, n, prf)
        meta["cases"].append(dict(i=i, B=B, rate=rate, bias=bias, tp=tp, pp=pp, p=p, n=n, ap=ap, prf=prf))
    print("G6 ok", len(meta["cases"]))
    save("G6_metrics", {}, meta)


# =========================================================================== G7
def g7():
    arrays, meta = {}, {"cases": []}
    for name, ctor, kw in (("adam", torch.optim.Adam, dict(lr=1e-3, weight_decay=1e-2)),
                           ("rmsprop", torch.optim.RMSprop, dict(lr=3e-4, weight_decay=5e-3))):
        p0 = dg.uniform(f"g7/{name}/p", (257,), -1, 1)
        p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = ctor([p], **kw)
        traj = []
        st = {}
        for step in range(1, 4):
            g = dg.uniform(f"g7/{name}/g{step}", (257,), -1, 1)
            p.grad = torch.from_numpy(g.copy())
            opt.step()
            traj.append(p.detach().numpy().copy())
        arrays[name + "_p3"] = traj[-1]
        arrays[name + "_p1"] = traj[0]
        # oracle
        po = p0.copy()
        if name == "adam":
            m = np.zeros_like(po); v = np.zeros_like(po)
            for step in range(1, 4):
                g = dg.uniform(f"g7/{name}/g{step}", (257,), -1, 1)
                po, m, v = orc.adam_step(po, g, m, v, step, kw["lr"], kw["weight_decay"])
        else:
            sq = np.zeros_like(po)
            for step in range(1, 4):
                g = dg.uniform(f"g7/{name}/g{step}", (257,), -1, 1)
                po, sq = orc.rmsprop_step(po, g, sq, kw["lr"], kw["weight_decay"])
        err = np.abs(po - traj[-1]).max()
        assert err < 1e-14, (name, err)
        meta["cases"].append(dict(name=name, **kw, n=257, steps=3, oracle_err=float(err)))
        print("G7", name, err)
    save("G7_optimizer_steps", arrays, meta)


# =========================================================================== G8
def g8():
    meta = {"cases": []}
    net = EmbraceNet("cpu", [4, 8], 16).double()
    try:
        net([torch.zeros(2, 4, dtype=torch.float64)])
    except AssertionError as e:
        meta["cases"].append(dict(name="modality_count", exc="AssertionError"))
    try:
        net([torch.zeros(2, 4, dtype=torch.float64), torch.zeros(2, 8, dtype=torch.float64)],
            availabilities=torch.tensor([[1.0, 0.0], [0.0, 1.0]]),
            selection_probabilities=torch.tensor([[0.0, 1.0], [0.0, 1.0]]))
    except RuntimeError as e:
        meta["cases"].append(dict(name="zero_distribution", exc="RuntimeError", msg=str(e)[:200]))
        try:
            orc.selection_cdf(np.array([[0.0, 1.0], [0.0, 1.0]]), np.array([[1.0, 0.0], [0.0, 1.0]]))
            raise SystemExit("oracle did not raise")
        except RuntimeError:
            pass
    assert len(meta["cases"]) == 2, meta
    print("G8", meta)
    save("G8_error_cases", {}, meta)


# =========================================================================== G9
def g9():
    """Reference fit_multimodal (training_models_multimodal.py:40) on in-memory lists:
    scores per epoch + final parameters pin loss, backward, optimizer and metric semantics."""
    arrays, meta = {}, {"cases": []}
    for opt_name, ctor, kw in (("adam", torch.optim.Adam, dict(lr=1e-3, weight_decay=1e-3)),
                               ("rmsprop", torch.optim.RMSprop, dict(lr=2e-4, weight_decay=1e-2))):
        tag = "g9/small"
        model, oracle, trial, hp, F_in = build_ref_model("small", tag)
        n_train, n_test, B = 5, 2, 64
        tr = [batch(f"{tag}/train{i}", B, F_in, 0.3) for i in range(n_train)]
        te = [batch(f"{tag}/test{i}", 2 * B, F_in, 0.3) for i in range(n_test)]
        mk = lambda bs: {"FFNN": [(a, y) for a, b, y in bs], "CNN": [(b, y) for a, b, y in bs]}
        opt = ctor(model.parameters(), **kw)
        seed, epochs = 9001, 2
        with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(seed)
            res = fit_multimodal(model, mk(tr), mk(te), "cpu", "A549", "active_E_vs_inactive_E", optimizer=opt,
                                 num_epochs=epochs, patience=4, verbose=False,
                                 checkpoint_path=os.path.join(td, "ckpt.pt"))
        # oracle replay of the same loop
        opt_o = ctor(oracle.parameters(), **kw)
        torch.manual_seed(seed)
        tr_scores, te_scores, prf_scores, losses = [], [], [], []
        for ep in range(epochs):
            oracle.train()
            s = 0.0
            for a, b, y in tr:
                l, ap = ref_step.train_step(oracle, opt_o, a, b, y)
                losses.append(l); s += ap
            tr_scores.append(s / n_train)
            oracle.eval()
            s = 0.0; prf = np.zeros(3)
            for a, b, y in te:
                l, ap, out = ref_step.eval_step(oracle, a, b, y)
                s += ap
                prf += orc.macro_prf(*orc.confusion_counts(out.detach().numpy(), y.numpy()))
            te_scores.append(s / n_test); prf_scores.append(prf / n_test)
        assert np.allclose(res[0], tr_scores, atol=1e-12) and np.allclose(res[1], te_scores, atol=1e-12), (res, tr_scores)
        assert np.allclose(np.array(res[2]), np.array(prf_scores), atol=1e-12)
        sd = model.state_dict()
        worst = 0.0
        chks = {}
        for k in oracle.names:
            d = (oracle.tensor(k).detach() - sd[k]).abs().max().item()
            worst = max(worst, d)
            chks[k] = dg.checksum(sd[k].numpy())
        assert worst < 1e-10, worst
        arrays[opt_name + "_dock0_w"] = sd["embracenet.docking_0.weight"].numpy()
        arrays[opt_name + "_post_last_w"] = sd[f"post.{3*hp['n_post_layers']}.weight"].numpy()
        arrays[opt_name + "_losses"] = np.array(losses)
        meta["cases"].append(dict(opt=opt_name, **kw, seed=seed, epochs=epochs, n_train=n_train, n_test=n_test, B=B,
                                  tag=tag, AUPRC_train=[float(x) for x in res[0]],
                                  AUPRC_test=[float(x) for x in res[1]],
                                  PRF_test=[[float(v) for v in x] for x in res[2]], final=chks,
                                  oracle_err=worst))
        print("G9", opt_name, "oracle max param diff", worst, "AUPRC", res[0], res[1])
    save("G9_fit_trajectory", arrays, meta)


# =========================================================================== G10
def g10():
    """Inference twin: the reference's EmbraceNetMultimodal_NoTrain rebuilt from a checkpoint in the
    harness format ({'model_state_dict', 'model_params'}, Kfold_CV_Multimodal :639-641) and called the way
    visual.Compare_Models_Result.get_model_predictions does (:266-293): .double(), eval, one region per call."""
    from BIOINF_tesi.models.EmbraceNetMultimodal_NoTrain import EmbraceNetMultimodal_NoTrain
    arrays, meta = {}, {"cases": []}
    cwd = os.getcwd()
    for cfg_name, N, seed in (("small", 12, 3101), ("post2", 8, 3102)):
        tag = f"g10/{cfg_name}"
        model, oracle, trial, hp, F_in = build_ref_model(cfg_name, tag)
        cell, task, n_iter = "A549", "active_E_vs_inactive_E", 1
        x1, x2, _ = batch(f"{tag}/N{N}", N, F_in)
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                torch.save({"model_state_dict": model.state_dict(), "model_params": dict(hp)},
                           f"{cell}_EmbraceNetMultimodal_{task}_{n_iter}_test_.pt")
                with contextlib.redirect_stderr(io.StringIO()):
                    twin = EmbraceNetMultimodal_NoTrain(cell, task, n_iter, F_in, device="cpu")
                state = torch.load(f"{cell}_EmbraceNetMultimodal_{task}_{n_iter}_test_.pt", map_location="cpu")
                twin.load_state_dict(state["model_state_dict"])
                twin.double().to("cpu")
                twin.eval()
                import warnings
                with torch.no_grad(), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    torch.manual_seed(seed)
                    per_sample = torch.stack([twin([x1[i:i + 1], x2[i:i + 1]]) for i in range(N)])     # [N, 2]
                    torch.manual_seed(seed + 1)
                    batched = twin([x1, x2])                                                          # [2N]
            finally:
                os.chdir(cwd)
        assert per_sample.shape == (N, 2) and batched.shape == (2 * N,)
        assert torch.allclose(per_sample.sum(1), torch.ones(N, dtype=torch.float64), atol=1e-12)
        key = f"{cfg_name}_N{N}"
        arrays[key + "_per_sample"] = per_sample.numpy()
        arrays[key + "_batched"] = batched.numpy()
        meta["cases"].append(dict(key=key, cfg=cfg_name, tag=tag, N=N, seed=seed, cell_line=cell, task=task, n_iter=n_iter,
                                  state_keys=list(model.state_dict().keys())))
        print("G10", key, "p1 range", float(per_sample[:, 1].min()), float(per_sample[:, 1].max()))
    save("G10_inference_twin", arrays, meta)


def g11():
    """Balanced batch index lists: the reference's BalancePos_BatchSampler (dataprepare.py:417-453) over
    label vectors from oracle/datagen.py, three epochs each (the sampler keeps its lists shuffled between epochs).  The
    sampler only reads pos_index / neg_index / len() of its data set, so a bare stand-in carries the labels (the
    reference's Dataset_Wrap needs a scikit-learn older than the one installed here)."""
    from BIOINF_tesi.data_pipe.dataprepare import BalancePos_BatchSampler

    class Labels:
        def __init__(self, y):
            self.pos_index = np.flatnonzero(y == 1).tolist()
            self.neg_index = np.flatnonzero(y == 0).tolist()
            self.n = len(y)

        def __len__(self):
            return self.n

    arrays, meta = {}, {"cases": []}
    for i, (n, rate, bs, seed) in enumerate([(23, 0.25, 5, 123), (1000, 0.1, 64, 123), (640, 0.3, 64, 7), (37, 0.5, 100, 123),
                                             (300, 0.02, 32, 123), (12, 0.0, 4, 123)]):
        y = dg.labels(f"g11/{i}/y", n, rate).reshape(-1) if rate > 0 else np.zeros(n, dtype=np.int64)
        sampler = BalancePos_BatchSampler(Labels(y), bs, random_state=seed)
        sizes, flat = [], []
        for _ in range(3):
            batches = list(iter(sampler))
            sizes.append([len(b) for b in batches])
            flat += [int(v) for b in batches for v in b]
        arrays[f"c{i}_idx"] = np.asarray(flat, dtype=np.int64)
        meta["cases"].append(dict(n=n, rate=rate, batch_size=bs, seed=seed, len=len(sampler), sizes=sizes,
                                  positives=int(y.sum())))
    save("G11_balanced_batches", arrays, meta)


# =========================================================================== G12
def g12():
    """EmbraceNet(bypass_docking=True) of the imported reference (EmbraceNetMultimodal.py:54-55): forward and the input
    gradients of sum(out * dout)."""
    arrays, meta = {}, {"cases": []}
    for (B, c) in [(8, 32), (64, 512), (37, 767), (130, 1024)]:
        for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            for av_kind, p_kind in (("none", "none"), ("mixed", "given"), ("none", "given")):
                case = f"g12/B{B}_c{c}"
                tag = f"{case}/{dt_name}/{av_kind}/{p_kind}"
                seed = 5000 + len(meta["cases"])
                X = [dg.uniform(case + "/x0", (B, c), -1, 1), dg.uniform(case + "/x1", (B, c), -1, 1)]
                dout = dg.uniform(case + "/dout", (B, c), -1, 1)
                avail = avail_variant(av_kind, case, B)
                p = None if p_kind == "none" else dg.uniform(case + "/p", (B, 2), 0.05, 1.0).astype(np.float32)
                net = EmbraceNet("cpu", [c, c], c, bypass_docking=True)
                assert len(list(net.parameters())) == 0
                xs = [torch.from_numpy(x).to(dt).requires_grad_(True) for x in X]
                torch.manual_seed(seed)
                out = net(xs, availabilities=None if avail is None else torch.from_numpy(avail),
                          selection_probabilities=None if p is None else torch.from_numpy(p))
                (out * torch.from_numpy(dout).to(dt)).sum().backward()
                torch.manual_seed(seed)
                u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
                cdf = orc.selection_cdf(np.ones((B, 2), np.float32) if p is None else p, avail)
                idx = orc.embrace_indices(cdf, u)
                npdt = np.float64 if dt_name == "f64" else np.float32
                Xc = [x.astype(npdt).astype(np.float64) for x in X]
                E = orc.embrace_bypass_forward(Xc, idx)
                dX = orc.embrace_bypass_backward(dout.astype(npdt).astype(np.float64), idx)
                ref = out.detach().double().numpy()
                assert np.array_equal(E, ref), tag                      # a select: exact
                for m in range(2):
                    assert np.array_equal(dX[m], xs[m].grad.double().numpy()), (tag, m)
                key = tag.replace("/", "_")
                arrays[key + "_idx"] = packbits(idx)
                meta["cases"].append(dict(tag=tag, key=key, case=case, B=B, c=c, dtype=dt_name, avail=av_kind, p=p_kind,
                                          seed=seed, out_chk=dg.checksum(ref), dx0_chk=dg.checksum(xs[0].grad.double().numpy()),
                                          dx1_chk=dg.checksum(xs[1].grad.double().numpy()), idx_ones=int(idx.sum())))
                print("G12", tag, "ones", int(idx.sum()))
    save("G12_bypass_docking", arrays, meta)


# =========================================================================== G13
def g13_inputs(case):
    name, B, ds, c, bypass = case["case"], case["B"], case["ds"], case["c"], case["bypass"]
    M = len(ds)
    X = [dg.uniform(f"{name}/x{m}", (B, d), -1, 1) for m, d in enumerate(ds)]
    W = [] if bypass else [dg.weight(f"{name}/w{m}", (c, d), d) for m, d in enumerate(ds)]
    b = [] if bypass else [dg.weight(f"{name}/b{m}", (c,), d) for m, d in enumerate(ds)]
    dout = dg.uniform(name + "/dout", (B, c), -1, 1)
    avail = None
    if case["avail"] == "mixed":
        a = dg.integers(name + "/avail", (B, M), 2).astype(np.float32)
        a[np.arange(B), dg.integers(name + "/avail_keep", (B,), M)] = 1.0       # at least one modality per row
        avail = a
    p = None if case["p"] == "none" else dg.uniform(name + "/p", (B, M), 0.05, 1.0).astype(np.float32)
    return X, W, b, dout, avail, p


def g13():
    """EmbraceNet of the imported reference with M != 2 modalities (EmbraceNetMultimodal.py:46-48), with and without
    docking layers: indices, outputs, gradients of sum(out * dout)."""
    arrays, meta = {}, {"cases": []}
    shapes = [(16, [8, 12, 20], 32, False), (64, [16, 64, 256, 100], 512, False), (37, [5], 30, False),
              (37, [30, 30, 30], 30, True), (33, [64] * 8, 64, True)]
    for (B, ds, c, bypass) in shapes:
        for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            for av_kind, p_kind in (("none", "none"), ("mixed", "given"), ("none", "given")):
                M = len(ds)
                case = dict(case=f"g13/B{B}_M{M}_c{c}_{int(bypass)}", B=B, ds=ds, c=c, bypass=bypass, avail=av_kind, p=p_kind)
                tag = f"{case['case']}/{dt_name}/{av_kind}/{p_kind}"
                seed = 7000 + len(meta["cases"])
                X, W, b, dout, avail, p = g13_inputs(case)
                net = EmbraceNet("cpu", ds, c, bypass_docking=bypass).to(dt)
                if not bypass:
                    with torch.no_grad():
                        for m in range(M):
                            getattr(net, f"docking_{m}").weight.copy_(torch.from_numpy(W[m]).to(dt))
                            getattr(net, f"docking_{m}").bias.copy_(torch.from_numpy(b[m]).to(dt))
                xs = [torch.from_numpy(x).to(dt).requires_grad_(True) for x in X]
                torch.manual_seed(seed)
                out = net(xs, availabilities=None if avail is None else torch.from_numpy(avail),
                          selection_probabilities=None if p is None else torch.from_numpy(p))
                (out * torch.from_numpy(dout).to(dt)).sum().backward()
                torch.manual_seed(seed)
                u = torch.rand(B * c, dtype=torch.float64).view(B, c).numpy()
                cdf = orc.selection_cdf(np.ones((B, M), np.float32) if p is None else p, avail)
                idx = orc.embrace_indices(cdf, u)
                npdt = np.float64 if dt_name == "f64" else np.float32
                Xc = [x.astype(npdt).astype(np.float64) for x in X]
                dE = dout.astype(npdt).astype(np.float64)
                ref = out.detach().double().numpy()
                tol = 1e-12 if dt_name == "f64" else 2e-5
                if bypass:
                    E = orc.embrace_bypass_forward(Xc, idx)
                    dX = orc.embrace_bypass_backward(dE, idx, M)
                    assert np.array_equal(E, ref), tag
                    D = Xc
                else:
                    Wc = [w.astype(npdt).astype(np.float64) for w in W]
                    bc = [v.astype(npdt).astype(np.float64) for v in b]
                    E, Z = orc.embrace_forward(Xc, Wc, bc, idx)
                    dX, dW, db = orc.embrace_backward(dE, Xc, Wc, Z, idx)
                    assert np.abs(E - ref).max() < tol, (tag, np.abs(E - ref).max())
                    D = [np.maximum(z, 0) for z in Z]
                    for m in range(M):
                        lin = getattr(net, f"docking_{m}")
                        assert np.abs(dW[m] - lin.weight.grad.double().numpy()).max() < tol * 10, (tag, m)
                        assert np.abs(db[m] - lin.bias.grad.double().numpy()).max() < tol * 10, (tag, m)
                for m in range(M):
                    assert np.abs(dX[m] - xs[m].grad.double().numpy()).max() < tol * 10, (tag, m)
                # the reference never returns idx: wherever the outputs tell the modalities apart it must be the oracle's
                pick = np.stack([np.abs(ref - d) for d in D], -1)
                ref_idx = pick.argmin(-1)
                srt = np.sort(pick, -1)
                amb = (srt[..., 1] - srt[..., 0] < 1e-6) if M > 1 else np.zeros_like(idx, bool)
                assert np.all((ref_idx == idx) | amb), tag
                key = tag.replace("/", "_")
                arrays[key + "_idx"] = idx.astype(np.uint8)
                arrays[key + "_out"] = ref.astype(np.float32) if B * c > 4096 else ref
                cm = dict(case, tag=tag, key=key, dtype=dt_name, seed=seed, out_chk=dg.checksum(ref),
                          dx_chk=[dg.checksum(x.grad.double().numpy()) for x in xs], hist=np.bincount(idx.ravel(), minlength=M).tolist())
                if not bypass:
                    cm["dw_chk"] = [dg.checksum(getattr(net, f"docking_{m}").weight.grad.double().numpy()) for m in range(M)]
                    cm["db_chk"] = [dg.checksum(getattr(net, f"docking_{m}").bias.grad.double().numpy()) for m in range(M)]
                meta["cases"].append(cm)
                print("G13", tag, "hist", cm["hist"])
    save("G13_m_modalities", arrays, meta)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13"]
    for w in which:
        globals()[w]()
