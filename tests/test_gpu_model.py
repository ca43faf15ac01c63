"""GPU parity of the whole drop-in model / train step against the golden fixtures generated from the imported
reference (G2 eval logits, G3 train step with gradients, G9 fit trajectory) -- `pytest -m gpu`.

Host-RNG ("parity") mode: the module draws the reference's random numbers from torch's CPU generator in the
reference's order and injects them, so the index tensor must be bit-exact and everything else agrees to
rounding.  Tolerances: fp64 models 1e-9 (logits) / 1e-8 relative (gradients through the fp32 loss: 1e-5 relative; trained parameters 1e-6);
fp32 models 1e-5 on logits (the north-star bar).
"""
import copy

import numpy as np
import pytest
import torch

from helpers import Golden, model_batch, model_fill, stock_prenets, unpack_idx
from oracle import datagen as dg
from oracle.configs import CONFIGS, FixedTrial

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(ea, cfg_name, tag, dtype=torch.float64):
    hp, F_in = CONFIGS[cfg_name]
    trial = FixedTrial(hp)
    model = ea.EmbraceNetMultimodal(trial, cell_line="A549", task="active_E_vs_inactive_E", device=DEV,
                                    in_features_FFNN=F_in)
    fill = model_fill(tag)
    model = model.double()
    with torch.no_grad():
        for key, t in model.state_dict().items():
            if "running_" in key or "num_batches" in key:
                continue
            t.copy_(torch.from_numpy(fill(key, tuple(t.shape))))
    model = model.to(dtype).to(DEV).set_rng("host")
    return model, trial, hp, F_in


def batch(tag, B, F_in, dtype, rate=0.1):
    x1, x2, y = model_batch(tag, B, F_in, rate)
    return torch.from_numpy(x1).to(DEV, dtype), torch.from_numpy(x2).to(DEV, dtype), torch.from_numpy(y).to(DEV)


@pytest.mark.parametrize("i", range(4))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_g2_eval_logits(ea, i, dtype):
    g = Golden("G2_model_eval_logits")
    case = g.meta["cases"][i]
    model, trial, hp, F_in = build(ea, case["cfg"], case["tag"], dtype)
    assert trial.calls == case["trial_calls"], "Optuna trial call order differs from the reference"
    assert sum(p.numel() for p in model.parameters()) == case["n_params"]
    x1, x2, _ = batch(f"{case['tag']}/B{case['B']}", case["B"], F_in, dtype)
    model.eval()
    torch.manual_seed(case["seed"])
    out = model([x1, x2])
    idx = model.embracenet.modality_indices().cpu().numpy()
    assert np.array_equal(idx, unpack_idx(g[case["key"] + "_idx"], case["B"], hp["EMBRACENET_embracement_size"]))
    err = np.abs(out.detach().double().cpu().numpy() - g[case["key"] + "_logits"]).max()
    assert err < (1e-9 if dtype == torch.float64 else 1e-5), err


@pytest.mark.parametrize("i", range(4))
def test_g3_train_step_gradients(ea, i):
    g = Golden("G3_model_train_step")
    case = g.meta["cases"][i]
    key = case["key"]
    model, trial, hp, F_in = build(ea, case["cfg"], case["tag"])
    x1, x2, y = batch(f"{case['tag']}/B{case['B']}", case["B"], F_in, torch.float64)
    model.train()
    cap = {}

    def hook(mod, args, kwargs):
        cap["in"] = list(args[0])
        for t in cap["in"]:
            t.retain_grad()
    h = model.embracenet.register_forward_pre_hook(hook, with_kwargs=True)
    torch.manual_seed(case["seed"])
    out = model([x1, x2], is_training=True)
    h.remove()
    c = hp["EMBRACENET_embracement_size"]
    idx = model.embracenet.modality_indices().cpu().numpy()
    assert np.array_equal(idx, unpack_idx(g[key + "_idx"], case["B"], c))
    if case["dropped"]:     # every row used exactly the modality the reference drew for it
        t = g[key + "_t"].astype(np.int64)
        assert np.array_equal(idx, np.repeat(t[:, None], c, 1))
    assert np.abs(out.detach().cpu().numpy() - g[key + "_logits"]).max() < 1e-9
    loss = ea.functional.weighted_ce(out, y)
    assert abs(loss.item() - case["loss"]) < 2e-6     # the reference's loss is fp32
    loss.backward()
    params = dict(model.named_parameters())
    for name, chk in case["grads"].items():
        got = dg.checksum(params[name].grad.cpu().numpy())
        scale = max(chk["abs"], 1e-6)   # floor: conv biases in front of BatchNorm have a zero gradient (noise)
        # the fp32 loss kernel rounds d(loss)/d(logits) to fp32, as the reference's fp32 criterion does
        assert abs(got["sum"] - chk["sum"]) < 1e-5 * scale and abs(got["dot"] - chk["dot"]) < 1e-5 * scale, name
        assert abs(got["abs"] - chk["abs"]) < 1e-5 * scale, name
    for nm, arr in (("embracenet.docking_0.weight", "_g_dock0_w"), ("embracenet.docking_0.bias", "_g_dock0_b"),
                    ("embracenet.docking_1.bias", "_g_dock1_b")):
        ref = g[key + arr]
        assert np.abs(params[nm].grad.cpu().numpy() - ref).max() < 2e-6 * max(1e-6, np.abs(ref).max()), nm
    for m, suffix in ((0, "_dX0"), (1, "_dX1")):
        ref = g[key + suffix].astype(np.float64)
        assert np.abs(cap["in"][m].grad.cpu().numpy() - ref).max() < 5e-6 * max(1e-9, np.abs(ref).max())
    sd = model.state_dict()
    for k in case["bn_keys"]:
        assert np.abs(sd[k].cpu().numpy() - g[key + "_bn_" + k.replace(".", "_")]).max() < 1e-10, k


@pytest.mark.parametrize("i", range(2))
def test_g9_fit_trajectory_matches_reference(ea, i, tmp_path):
    from embracenet_amd import optim, training
    g = Golden("G9_fit_trajectory")
    case = g.meta["cases"][i]
    model, trial, hp, F_in = build(ea, "small", case["tag"])
    B, n_train, n_test = case["B"], case["n_train"], case["n_test"]
    def mk(prefix, n, bs):
        bt = [model_batch(f"{case['tag']}/{prefix}{k}", bs, F_in, 0.3) for k in range(n)]
        t = lambda a: torch.from_numpy(a)
        return {"FFNN": [(t(a), t(y)) for a, b, y in bt], "CNN": [(t(b), t(y)) for a, b, y in bt]}
    cls = optim.Adam if case["opt"] == "adam" else optim.RMSprop
    opt = cls(model.parameters(), lr=case["lr"], weight_decay=case["weight_decay"])
    torch.manual_seed(case["seed"])
    res = training.fit_multimodal(model, mk("train", n_train, B), mk("test", n_test, 2 * B), DEV, "A549",
                                  "active_E_vs_inactive_E", optimizer=opt, num_epochs=case["epochs"], patience=4,
                                  verbose=False, checkpoint_path=str(tmp_path / "ck.pt"), precision="float64")
    assert np.allclose(res[0], case["AUPRC_train"], atol=1e-12), (res[0], case["AUPRC_train"])
    assert np.allclose(res[1], case["AUPRC_test"], atol=1e-12)
    assert np.allclose(np.array(res[2]), np.array(case["PRF_test"]), atol=1e-12)
    sd = model.state_dict()
    for name, chk in case["final"].items():
        got = dg.checksum(sd[name].cpu().numpy())
        assert abs(got["sum"] - chk["sum"]) < 1e-6 * max(chk["abs"], 1e-12), name
    w = sd["embracenet.docking_0.weight"].cpu().numpy()
    assert np.abs(w - g[case["opt"] + "_dock0_w"]).max() < 1e-6
    # resumability: a second call with the same checkpoint path reloads scores instead of training (:95-100)
    res2 = training.fit_multimodal(model, mk("train", 1, B), mk("test", 1, B), DEV, "A549", "active_E_vs_inactive_E",
                                   optimizer=opt, num_epochs=1, checkpoint_path=str(tmp_path / "ck.pt"))
    assert res2[0] == res[0]


def test_model_is_picklable_and_resettable(ea, tmp_path):
    from embracenet_amd import metrics
    model, trial, hp, F_in = build(ea, "small", "pk")
    x1, x2, _ = batch("pk/B16", 16, F_in, torch.float64)
    model.eval()
    model([x1, x2])
    torch.save(model, tmp_path / "m.pt")                       # whole-object save, as the harness does (:413)
    m2 = torch.load(tmp_path / "m.pt", weights_only=False)
    m2.set_rng("host")
    torch.manual_seed(3); a = model([x1, x2])
    torch.manual_seed(3); b = m2([x1, x2])
    assert torch.equal(a, b)
    before = model.embracenet.docking_1.weight.clone()
    model.apply(metrics.weight_reset)
    assert not torch.equal(before, model.embracenet.docking_1.weight)


def test_philox_training_step_bf16_runs_and_learns(ea):
    """perf-mode smoke: bf16 compute, device-side RNG, fused Adam; the loss must go down on a fixed batch."""
    from embracenet_amd import optim, training
    model, trial, hp, F_in = build(ea, "small", "bf", torch.float32)
    model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=7)
    x1, x2, y = batch("bf/B256", 256, F_in, torch.float32, rate=0.3)
    y = (x1[:, 0] > 0.5).long().view(-1, 1)                    # learnable labels
    opt = optim.Adam(model.parameters(), lr=2e-3)
    model.train()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        out = model([x1, x2], is_training=True)
        loss = ea.functional.weighted_ce(out, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.02, losses


@pytest.mark.parametrize("mode", ["autograd", "runner"])
def test_prenet_riders_change_nothing(ea, mode):
    """The epigenomic MLP launches riding on kernels of the sequence CNN (csrc/rider.h, model.ride_prenets) are the same
    computations issued inside other launches: losses, parameters and BatchNorm statistics after a few bf16 steps are
    bit-identical to plain launches -- with a step runner (slab reductions deferred: forward AND backward ride) and with a
    plain autograd loop (forward rides; the backward is launched on its own because its reduction is not deferred)."""
    from embracenet_amd import optim, training
    def run(ride):
        model, trial, hp, F_in = build(ea, "cfg1", "rd", torch.float32)
        model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=5)
        model.ride_prenets = ride
        opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
        model.train()
        losses = []
        if mode == "runner":
            runner = training.StepRunner(model, opt, DEV)
            table = ea.metrics.StepTable(4, DEV)
        for k in range(4):
            a, b, y = model_batch(f"rd/{k}", 96 if k < 3 else 40, F_in, 0.3)        # (the last batch: partial row blocks)
            x1, x2, yy = torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y)
            if mode == "runner":
                runner.train_step(x1, x2, yy, table)
            else:
                opt.zero_grad()
                loss = ea.functional.weighted_ce(model([x1.to(DEV), x2.to(DEV)], is_training=True), yy.to(DEV))
                loss.backward()
                opt.step()
                losses.append(loss.item())
        if mode == "runner":
            losses = table.fetch()[0].tolist()
        return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    (la, sa), (lb, sb) = run(False), run(True)
    assert la == lb, (la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("precision", ["bfloat16", "float32"])
def test_graph_replayed_steps_equal_eager_steps(ea, precision):
    """training.set_graph_steps: fit_multimodal replaying captured train / eval steps (one hipGraph per batch shape, ragged
    last batch included) ends in bit-identical parameters, BatchNorm statistics and per-epoch scores as the eager loop."""
    from embracenet_amd import optim, training
    def run(graph):
        model, trial, hp, F_in = build(ea, "small", "gr", torch.float32)
        model.set_rng("philox", seed=11)
        def mk(prefix, sizes):
            bt = [model_batch(f"gr/{prefix}{k}", bs, F_in, 0.3) for k, bs in enumerate(sizes)]
            t = lambda a: torch.from_numpy(a).float()
            return {"FFNN": [(t(a), torch.from_numpy(y)) for a, b, y in bt], "CNN": [(t(b), torch.from_numpy(y)) for a, b, y in bt]}
        opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        training.set_graph_steps(graph)
        try:
            res = training.fit_multimodal(model, mk("train", [64] * 5 + [24]), mk("test", [128] * 4), DEV, "A549",
                                          "active_E_vs_inactive_E", optimizer=opt, num_epochs=3, patience=10,
                                          verbose=False, precision=precision)
        finally:
            training.set_graph_steps(False)
        return res, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    (ra, sa), (rb, sb) = run(False), run(True)
    assert ra[0] == rb[0] and ra[1] == rb[1]
    assert np.array_equal(np.array(ra[2]), np.array(rb[2]))
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("i", range(2))
def test_g10_inference_twin_matches_reference(ea, i, tmp_path):
    """Row f3: EmbraceNetMultimodal_NoTrain rebuilt from a harness-format checkpoint reproduces the reference's twin --
    per-region calls as visual.Compare_Models_Result makes them, and one batched call (fp64, host RNG replay)."""
    g = Golden("G10_inference_twin")
    case = g.meta["cases"][i]
    src, trial, hp, F_in = build(ea, case["cfg"], case["tag"])
    N = case["N"]
    ck = tmp_path / f"{case['cell_line']}_EmbraceNetMultimodal_{case['task']}_{case['n_iter']}_test_.pt"
    torch.save({"model_state_dict": {k: v.cpu() for k, v in src.state_dict().items()}, "model_params": dict(hp)}, ck)
    assert list(src.state_dict().keys()) == case["state_keys"]
    twin = ea.EmbraceNetMultimodal_NoTrain(case["cell_line"], case["task"], case["n_iter"], F_in, device=DEV,
                                           checkpoint_dir=str(tmp_path))
    state = torch.load(ck, map_location=DEV, weights_only=True)
    twin.load_state_dict(state["model_state_dict"])
    twin.double().to(DEV)
    twin.set_rng("host")
    twin.eval()
    assert all(not p.requires_grad for p in twin.FFNN.parameters()) and all(not p.requires_grad for p in twin.CNN.parameters())
    x1, x2, _ = batch(f"{case['tag']}/N{N}", N, F_in, torch.float64)
    with torch.no_grad():
        torch.manual_seed(case["seed"])
        per = torch.stack([twin([x1[j:j + 1], x2[j:j + 1]]) for j in range(N)])
        torch.manual_seed(case["seed"] + 1)
        batched = twin([x1, x2])
    assert per.shape == (N, 2) and batched.shape == (2 * N,)
    assert np.abs(per.cpu().numpy() - g[case["key"] + "_per_sample"]).max() < 1e-9
    assert np.abs(batched.cpu().numpy() - g[case["key"] + "_batched"]).max() < 1e-9
    # the batched entry point returns the positive-class column
    torch.manual_seed(case["seed"] + 1)
    p1 = twin.predict_proba(x1.cpu(), x2.cpu(), batch_size=N)
    assert np.abs(p1.cpu().numpy() - g[case["key"] + "_batched"].reshape(N, 2)[:, 1]).max() < 1e-9


class _RandomTrial:
    """Draws every hyper-parameter the model asks for uniformly from the range / choices it offers (the reference's Optuna
    search space, whatever it is: FFNN depth and widths, CNN depth, channels, kernel sizes, dropouts, embracement size,
    post stack)."""

    def __init__(self, seed):
        self.rng, self.asked = np.random.RandomState(seed), {}

    def suggest_int(self, name, lo, hi):
        self.asked[name] = int(self.rng.randint(lo, hi + 1))
        return self.asked[name]

    def suggest_categorical(self, name, choices):
        self.asked[name] = choices[int(self.rng.randint(len(choices)))]
        return self.asked[name]

    def suggest_float(self, name, lo, hi):
        self.asked[name] = float(self.rng.uniform(lo, hi))
        return self.asked[name]


@pytest.mark.parametrize("seed", range(24))
def test_random_points_of_the_search_space(ea, seed):
    """Shape coverage: two dozen random architectures of the reference's search space.  (1) fp64 eval logits of the HIP path
    equal the same module run on stock torch operators (helpers.stock_prenets for the pre-nets, fusion still HIP) to 1e-9;
    (2) fp32 and bf16 training steps through StepRunner (fused head where its shape rules allow, three launches otherwise)
    run, give finite losses and move the parameters."""
    from embracenet_amd import optim, training
    F_in = [48, 152, 429, 562][seed % 4]
    B = [32, 37, 64, 100][seed % 4]
    x1, x2, y = model_batch(f"rs/{seed}", B, F_in, 0.3)
    trial = _RandomTrial(100 + seed)
    model = ea.EmbraceNetMultimodal(trial, cell_line="A549", task="active_E_vs_inactive_E", device=DEV, in_features_FFNN=F_in)
    model = model.double().to(DEV).set_rng("host")
    model.eval()
    a, b = torch.from_numpy(x1).to(DEV), torch.from_numpy(x2).to(DEV)
    with torch.no_grad():
        torch.manual_seed(5); z_hip = model([a, b])
        with stock_prenets(model):
            torch.manual_seed(5); z_ref = model([a, b])
    assert (z_hip - z_ref).abs().max().item() < 1e-9 * max(1.0, z_ref.abs().max().item()), trial.asked
    for precision in ("float32", "bfloat16"):
        m = training.prepare_model(copy.deepcopy(model), DEV, precision).set_rng("philox", seed=seed)
        opt = optim.Adam(m.parameters(), lr=1e-3)
        runner = training.StepRunner(m, opt, DEV)
        table = ea.metrics.StepTable(4, DEV)
        before = m.embracenet.docking_0.weight.detach().clone()
        m.train()
        for _ in range(2):
            runner.train_step(torch.from_numpy(x1).float(), torch.from_numpy(x2).float(), torch.from_numpy(y), table)
        losses, counts = table.fetch()
        assert np.isfinite(losses).all() and (losses > 0).all(), (precision, trial.asked, losses)
        assert counts[:, 3].tolist() == [B, B] and counts[:, 2].tolist() == [int(y.sum())] * 2
        assert not torch.equal(before, m.embracenet.docking_0.weight), (precision, trial.asked)


@pytest.mark.parametrize("case", ["plain", "device_dropout", "availabilities"])
def test_inline_selection_equals_the_prepared_cdf_path(ea, case):
    """emb_embrace_fwd_select (row thresholds computed inside the forward launch) vs emb_select_prep + emb_embrace_fwd:
    same code bytes (selected modality, ReLU bit) and the same fused output, bit for bit, over several RNG steps."""
    F = ea.functional
    B, d0, d1, c = 77, 16, 200, 96
    g = torch.Generator(device="cpu").manual_seed(4)
    x0, x1 = torch.rand(B, d0, generator=g).to(DEV), torch.rand(B, d1, generator=g).to(DEV)
    w0, w1 = (torch.rand(c, d0, generator=g) - 0.5).to(DEV), ((torch.rand(c, d1, generator=g) - 0.5) * 0.2).to(DEV)
    b0, b1 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    p = torch.tensor([[0.58, 0.42]], device=DEV)
    avail = None
    if case == "availabilities":
        avail = torch.nn.functional.one_hot((torch.rand(B, generator=g) > 0.3).long(), 2).float().to(DEV)
        avail[::5] = 1.0
    for step in range(1, 5):
        rng = F.RngState(seed=9, step_val=step, row0=1000)
        cdf0, st = F.select_prep(p, avail, B, rng=rng, device_dropout=(case == "device_dropout"))
        E_a, code_a = F.embrace(x0, x1, w0, b0, w1, b1, cdf0, rng=rng)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        E_b, code_b = F.embrace(x0, x1, w0, b0, w1, b1, F.SelectInline(p, avail, case == "device_dropout", status), rng=rng)
        assert torch.equal(code_a, code_b) and torch.equal(E_a, E_b), (case, step)
        assert int(status) == int(st) == 0
    bad = torch.tensor([[0.5, -0.5]], device=DEV)               # invalid distribution: the sticky status bit is raised
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    F.embrace(x0, x1, w0, b0, w1, b1, F.SelectInline(bad, None, False, status), rng=F.RngState(seed=1, step_val=1))
    assert int(status) & ea.embracenet.STATUS_INVALID_DISTRIBUTION


def test_graph_steps_follow_a_changing_learning_rate(ea):
    """Launch arguments are frozen in a captured step, so the graphs are keyed on the optimizer's hyper-parameters: changing
    lr mid-run (what an lr scheduler does) must give the parameters of the eager loop, bit for bit."""
    from embracenet_amd import optim, training
    def run(graph):
        model, trial, hp, F_in = build(ea, "small", "lr", torch.float32)
        model.set_rng("philox", seed=21)
        model = training.prepare_model(model, DEV, "float32")
        opt = optim.Adam(model.parameters(), lr=1e-3)
        runner = training.StepRunner(model, opt, DEV, graph=graph)
        table = ea.metrics.StepTable(32, DEV)
        model.train()
        for k in range(14):
            if k in (5, 9):
                for g_ in opt.param_groups:
                    g_["lr"] *= 0.5
            a, b, y = model_batch(f"lr/{k % 3}", 48, F_in, 0.3)
            runner.train_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
        return table.fetch()[0], {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, len(runner._graphs)
    (la, sa, _), (lb, sb, n_graphs) = run(False), run(True)
    assert n_graphs == 3 and np.array_equal(la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("precision,opt_name", [("float64", "Adam"), ("float32", "Adam"), ("bfloat16", "Nadam"), ("float32", "RMSprop")])
def test_optimizer_launch_sums_the_gradient_slabs_itself(ea, precision, opt_name):
    """Single process: the backward kernels' per-slice gradient slabs are summed inside the multi-tensor optimizer launch
    (no reduction launch; csrc/loss_optim.hip) -- same parameters and the same .grad tensors as with the reduction launch in
    front of the optimizer, up to the summation order of the slices."""
    from embracenet_amd import training
    def run(consume):
        model, trial, hp, F_in = build(ea, "post2", "slabopt", torch.float64)
        model = training.prepare_model(model, DEV, precision).set_rng("philox", seed=3)
        opt = training.make_optimizer(opt_name, model.parameters(), 2e-3, 1e-3)
        runner = training.StepRunner(model, opt, DEV)
        runner.consume_slabs = consume
        table = ea.metrics.StepTable(8, DEV)
        model.train()
        cast = torch.float64 if precision == "float64" else torch.float32
        for k in range(3):
            a, b, y = model_batch(f"slabopt/{k}", 96, F_in, 0.3)
            runner.train_step(torch.from_numpy(a).to(cast), torch.from_numpy(b).to(cast), torch.from_numpy(y), table)
        losses, counts = table.fetch()
        return (losses, counts, {k: v.detach().double().cpu() for k, v in model.state_dict().items()},
                {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()})
    la, ca, sa, ga = run(False)
    lb, cb, sb, gb = run(True)
    tol = 1e-11 if precision == "float64" else 3e-5
    assert np.array_equal(ca, cb) and np.abs(la - lb).max() < 1e-5
    # a conv bias in front of BatchNorm has a mathematically zero gradient: what arrives is rounding noise, which the
    # normalising optimizers turn into +-lr steps -- not comparable between two summation orders
    noise = lambda k: k.startswith("CNN.CNN_model.") and k.endswith(".bias") and int(k.split(".")[2]) % 5 == 0
    for k in sa:
        if not noise(k):
            assert (sa[k] - sb[k]).abs().max().item() <= tol * max(1.0, sa[k].abs().max().item()), k
    for k in ga:
        if not noise(k):
            assert (ga[k] - gb[k]).abs().max().item() <= tol * max(1e-3, ga[k].abs().max().item()), ("grad", k)


def test_captured_steps_survive_workspace_growth_and_cache_release(ea):
    """A captured step graph has the scratch buffers' addresses baked in.  Larger batches arriving later (the test loader
    runs at twice the train batch, BalancePos batches differ by a row) must not hand those buffers back to the allocator:
    capture at B, run B + 1 and 2B eagerly (workspaces grow), release the cache, let new long-lived tensors take whatever
    was freed, replay the B graph -- parameters must equal the all-eager run bit for bit."""
    from embracenet_amd import optim, training
    def run(graph):
        model, trial, hp, F_in = build(ea, "small", "wsg", torch.float32)
        model.set_rng("philox", seed=31)
        model = training.prepare_model(model, DEV, "float32")
        opt = optim.Adam(model.parameters(), lr=1e-3)
        runner = training.StepRunner(model, opt, DEV, graph=graph)
        table = ea.metrics.StepTable(64, DEV)
        model.train()
        keep = []
        sizes = [48] * 4 + [49, 96] + [48] * 4                  # capture happens at the third 48-row step
        for k, bs in enumerate(sizes):
            a, b, y = model_batch(f"wsg/{bs}", bs, F_in, 0.3)
            runner.train_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
            if k == 5:                                          # after the larger batches: free what can be freed, then occupy it
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                keep = [torch.full((1 << 18,), 7.0, device=DEV) for _ in range(24)]
        torch.cuda.synchronize()
        assert all(bool((t == 7.0).all()) for t in keep), "a replayed graph wrote into memory it no longer owns"
        return table.fetch()[0], {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, len(runner._graphs)
    (la, sa, _), (lb, sb, n_graphs) = run(False), run(True)
    assert n_graphs >= 1 and np.array_equal(la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


# ------------------------------------------------------------------------- bf16 full model at BASELINE config 2
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_bf16_full_model_logits_at_cfg2_vs_oracle(ea, mode):
    """BASELINE configs[1] as benchmarked: cfg1's networks with c = 256, B = 1024, bf16 storage / fp32 accumulation.
    Oracle = oracle/ref_step.py in fp64 fed the SAME bf16-rounded parameters and inputs ("fp64 reference vs
    fp32/bf16 build").  Host-RNG mode, so the index tensor is bit-exact; logits agree to 3e-2 of the logit scale: the
    HIP path rounds six intermediate activations to bf16 (2^-9 relative each: three FFNN layers, two pooled conv blocks, E)
    on top of bf16 products accumulated in fp32 -- measured error is ~1e-2.  Train mode adds batch-statistics BatchNorm and
    the modality-dropout draws of EmbraceNetMultimodal.py:178-182."""
    from oracle import ref_step
    from oracle.configs import CFG1
    from embracenet_amd import training
    hp, F_in, B = dict(CFG1, EMBRACENET_embracement_size=256), 48, 1024
    bf = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(torch.bfloat16).double().numpy()
    fill = model_fill("bf16cfg2")
    rfill = lambda key, shape: bf(fill(key, shape))
    oracle_m = ref_step.OracleEmbraceNetMultimodal(hp, F_in)
    oracle_m.set_tensors(rfill)
    model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="A549", task="active_E_vs_inactive_E", device=DEV,
                                    in_features_FFNN=F_in).double()
    with torch.no_grad():
        for key, t in model.state_dict().items():
            if "running_" not in key and "num_batches" not in key:
                t.copy_(torch.from_numpy(rfill(key, tuple(t.shape))))
    model = training.prepare_model(model, DEV, "bfloat16").set_rng("host")
    x1, x2, _ = model_batch("bf16cfg2/B1024", B, F_in, 0.1)
    x1 = bf(x1)
    seeds = (3,) if mode == "eval" else (0, 1, 2, 3)        # train: cover both branches of the modality-dropout gate
    branches = set()
    for seed in seeds:
        if mode == "eval":
            oracle_m.eval(); model.eval()
        else:
            oracle_m.train(); model.train()
        torch.manual_seed(seed)
        with torch.no_grad():
            want = oracle_m([torch.from_numpy(x1), torch.from_numpy(x2)], is_training=(mode == "train")).numpy()
        branches.add(oracle_m.last["t"] is not None)
        torch.manual_seed(seed)
        with torch.no_grad():
            got = model([torch.from_numpy(x1).to(DEV, torch.bfloat16), torch.from_numpy(x2).to(DEV, torch.bfloat16)],
                        is_training=(mode == "train"))
        assert np.array_equal(model.embracenet.modality_indices().cpu().numpy(), oracle_m.last["idx"].numpy()), (mode, seed)
        got = got.double().cpu().numpy()
        scale = max(1.0, np.abs(want).max())
        err = np.abs(got - want).max() / scale
        assert err < 3e-2, (mode, seed, err)
        assert (np.argmax(got, 1) == np.argmax(want, 1)).mean() > 0.97
    if mode == "train":
        assert branches == {True, False}, "seeds no longer cover both modality-dropout branches"


# ------------------------------------------------------------------------- Param_Search_Multimodal.objective
class _ObjectiveTrial(FixedTrial):
    """FixedTrial plus the part of optuna's trial API that Param_Search_Multimodal.objective touches
    (training_models_multimodal.py:307-415): report / should_prune / number / log-uniform suggestions."""

    def __init__(self, params, number=7):
        super().__init__(params)
        self.number, self.reported = number, []

    def suggest_float(self, name, low, high, log=False):
        return self._get(name)

    def suggest_loguniform(self, name, low, high):
        return self._get(name)

    def report(self, value, step):
        self.reported.append((step, float(value)))

    def should_prune(self):
        return False


@pytest.mark.parametrize("opt_name", ["Nadam", "Adam", "RMSprop"])
def test_param_search_objective_with_a_fixed_trial(ea, opt_name, tmp_path):
    """One trial of the reference's search loop (training_models_multimodal.py:307-415) on device loaders: builds the model
    from the trial, draws optimizer / lr / weight_decay in the reference's order, trains with per-epoch report, saves the
    whole model.  Each of the three optimizers of the search space (:318-325) is exercised."""
    from embracenet_amd import data, training
    hp, F_in = CONFIGS["small"]
    mk = lambda tag, n: (dg.uniform(f"ps/{tag}/x1", (n, F_in)), dg.onehot_sequence(f"ps/{tag}/seq", n), dg.labels(f"ps/{tag}/y", n, 0.3))
    xtr, str_, ytr = mk("train", 160)
    xte, ste, yte = mk("test", 64)
    train = data.device_loaders(xtr, str_, ytr, 32, DEV, balanced=True, random_state=123, feature_dtype=torch.float32)
    test = data.device_loaders(xte, ste, yte, 64, DEV, balanced=False, random_state=153, feature_dtype=torch.float32)
    ps = training.Param_Search_Multimodal(ea.EmbraceNetMultimodal, train, test, num_epochs=2, study_name=str(tmp_path / "study_"),
                                          device=DEV, cell_line="A549", task="active_E_vs_inactive_E", precision="float32")
    trial = _ObjectiveTrial(dict(hp, optimizer=opt_name, lr=2e-3, weight_decay=1e-3))
    score = ps.objective(trial)
    n_model = len([c for c in trial.calls if c not in ("optimizer", "lr", "weight_decay")])
    assert trial.calls[n_model:] == ["optimizer", "lr", "weight_decay"], "optimizer draws must follow the model's (:318-321)"
    assert [s_ for s_, _ in trial.reported] == [1, 2] and trial.reported[-1][1] == score
    assert 0.0 <= score <= 1.0 and np.isfinite(score)
    saved = torch.load(str(tmp_path / "study_") + "7.pt", weights_only=False)
    assert type(saved).__name__ == "EmbraceNetMultimodal"
    for a, b in zip(saved.state_dict().values(), ps.model.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu())


def test_kfold_cv_driver_runs_on_device_loaders(ea, tmp_path, monkeypatch):
    """Kfold_CV_Multimodal.__call__ (training_models_multimodal.py:645-798) end to end on stubs of the reference's data
    pipeline object and of the Optuna study: folds, train / validation split, tuning, re-initialised best model, fit, scores.
    Nothing of the reference is imported."""
    import pandas as pd
    from sklearn.model_selection import KFold
    from embracenet_amd import training
    hp, F_in = CONFIGS["small"]
    N = 150
    X1 = pd.DataFrame(dg.uniform("kf/x1", (N, F_in)))
    bases = np.array(list("acgt"))[dg.integers("kf/seq", (N, 256), 4)]
    X2 = pd.DataFrame({"seq": ["".join(r) for r in bases]})
    y = pd.DataFrame({"y": dg.labels("kf/y", N, 0.35).reshape(-1)})

    class _DataClass:
        def return_index_data_for_cv(self, cell_line, sequence, n_folds, random_state):
            return KFold(n_splits=n_folds, shuffle=True, random_state=random_state), (X2 if sequence else X1), y

    class _Pipeline:
        data_class = _DataClass()

    def fake_run_trial(self):                                    # one "trial" instead of an Optuna study (optuna is not installed)
        trial = _ObjectiveTrial(dict(hp, optimizer="Adam", lr=1e-3, weight_decay=1e-3), number=0)
        self.objective(trial)
        self.best_model = torch.load(f"{self.study_name}0.pt", weights_only=False)
        self.best_params = dict(optimizer="Adam", lr=1e-3, weight_decay=1e-3)
    monkeypatch.setattr(training.Param_Search_Multimodal, "run_trial", fake_run_trial)
    monkeypatch.chdir(tmp_path)
    cv = training.Kfold_CV_Multimodal()
    cv(_Pipeline(), "A549", DEV, task="active_E_vs_inactive_E", model=ea.EmbraceNetMultimodal, n_folds=2, num_epochs=2,
       batch_size=16, study_name=str(tmp_path / "cv"), test_model_path="best", precision="float32")
    assert len(cv.scores_dict["final_test_AUPRC_scores"]) == 2 and 0.0 <= cv.scores_dict["average_CV_AUPRC"] <= 1.0
    assert (tmp_path / "models_" / "best.pt").exists()
    with pytest.raises(NotImplementedError):                     # a split the reference would re-balance: loud, not silent
        cv.rebalance_threshold = 0.6                             # positives / negatives = 0.35 / 0.65 = 0.54 < 0.6
        cv.build_dataloaders_forCV(X1, X2, y, 16, True, False)


def test_parked_first_block_finish_falls_back_when_the_optimizer_lacks_its_tensors(ea):
    """bf16 step runner: the first conv block's backward parks its per-channel finish for the optimizer launch
    (csrc/first_fin.h).  An optimizer that does not hold the block's beta cannot take it over: the launch must then run the
    finish the classic way first -- every parameter the optimizer does hold ends up exactly as in the all-parameters run
    (Adam is element-wise), and the left-out one keeps its value while its .grad is still produced."""
    from embracenet_amd import optim, training
    def run(leave_out):
        model, trial, hp, F_in = build(ea, "cfg1", "park", torch.float32)
        model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=9)
        names = dict(model.named_parameters())
        skip = [k for k in names if k.startswith("CNN.CNN_model.") and k.endswith(".bias") and "1." in k.split("CNN_model.")[1][:2]]
        assert len(skip) == 1, skip                       # BatchNorm beta of block 1 (CNN_pre.py:37-44: conv, bn, relu, pool)
        params = [p for k, p in names.items() if not (leave_out and k == skip[0])]
        opt = optim.Adam(params, lr=1e-3, weight_decay=1e-3)
        runner = training.StepRunner(model, opt, DEV)
        table = ea.metrics.StepTable(2, DEV)
        model.train()
        before = names[skip[0]].detach().clone()
        for k in range(1):
            a, b, y = model_batch(f"park/{k}", 64, F_in, 0.3)
            runner.train_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
        torch.cuda.synchronize()
        return ({k: v.detach().cpu().clone() for k, v in names.items()}, names[skip[0]].grad.detach().cpu().clone(),
                before.cpu(), skip[0], table.fetch()[0].tolist())
    pa, ga, _, name, la = run(False)
    pb, gb, before, _, lb = run(True)
    assert la[0] == lb[0]                                  # the step sees identical parameters
    assert torch.equal(pb[name], before)                   # not in the optimizer: untouched
    assert torch.isfinite(gb).all() and gb.abs().max() > 0 and torch.allclose(ga, gb, rtol=1e-5, atol=1e-7)
    for k in pa:
        if k == name:
            continue
        if k.startswith("CNN.CNN_model.0.") or k.startswith("CNN.CNN_model.1."):
            # the standalone finish sums its slab with 1024 threads, the one inside the optimizer launch with 256: another order.
            # The first Adam step moves every element by ~lr whatever the gradient's size, so an element whose gradient is
            # rounding noise around zero may move the other way
            assert (pa[k] - pb[k]).abs().max().item() <= 2.1e-3, k
            assert ((pa[k] - pb[k]).abs() > 1e-6).float().mean().item() < 0.01, k
        else:
            assert torch.equal(pa[k], pb[k]), k
