"""CPU: the C-ABI library loads and exports every symbol the header declares; host-side logic of the package
(no compute calls -- there is no GPU here and no CPU fallback to call)."""
import copy
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from helpers import Golden
from oracle.configs import CFG1, CFG1_F, CONFIGS, FixedTrial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(ea):
    header = open(os.path.join(ROOT, "include", "embrace_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(emb_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 14
    assert declared == set(ea._lib.SIGNATURES), "binding table and header disagree"
    if not os.path.exists(ea._lib.LIB_PATH):
        ea._lib.build()
    lib = ctypes.CDLL(ea._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert ea._lib.lib().emb_abi_version() == 2


def test_abi_rejects_bad_arguments_without_touching_a_gpu(ea):
    L = ea._lib.lib()
    assert L.emb_embrace_fwd(None, None, None, None, None, None, None, None, 0, 0, None, 0, None, None, 1, 1, 1, 1, 0, None) == -1
    assert b"null pointer" in L.emb_last_error()
    assert L.emb_cast(None, 0, None, 0, 4, None) == -1
    assert L.emb_select_prep(ctypes.c_void_p(16), 3, None, 0, 0, 0, None, 0, ctypes.c_void_p(16), ctypes.c_void_p(16), 8, None) == -1
    assert b"p_rows" in L.emb_last_error()


def test_cpu_tensors_fail_loudly(ea):
    net = ea.EmbraceNet("cpu", [4, 8], 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net([torch.zeros(2, 4), torch.zeros(2, 8)])


def test_model_construction_mirrors_reference(ea):
    g = Golden("G2_model_eval_logits")
    for case in g.meta["cases"]:
        hp, F_in = CONFIGS[case["cfg"]]
        trial = FixedTrial(hp)
        m = ea.EmbraceNetMultimodal(trial, cell_line="A549", task="active_E_vs_inactive_E", device="cpu",
                                    in_features_FFNN=F_in)
        assert trial.calls == case["trial_calls"]
        assert sum(p.numel() for p in m.parameters()) == case["n_params"]
        assert type(m).__name__ == "EmbraceNetMultimodal"
    m = ea.EmbraceNetMultimodal(FixedTrial(CFG1), cell_line="A549", task="t", device="cpu", in_features_FFNN=CFG1_F)
    keys = list(m.state_dict().keys())
    for k in ("FFNN.model.0.weight", "FFNN.model.6.bias", "CNN.CNN_model.0.weight", "CNN.CNN_model.1.running_mean",
              "CNN.CNN_model.6.weight", "embracenet.docking_0.weight", "embracenet.docking_1.bias", "post.0.weight"):
        assert k in keys, k
    assert not any("selection" in k for k in keys)                    # plain attribute, as in the reference (:157)
    assert tuple(m.embracenet.docking_1.weight.shape) == (512, 1856) and m.CNN_pre_output_size == 1856
    assert m.FFNN_pre_output_size == 16 and abs(float(m.selection_probabilities.sum()) - 1.0) < 1e-7
    m2 = pickle.loads(pickle.dumps(m)); m3 = copy.deepcopy(m)          # harness: deepcopy (:277), torch.save (:413)
    assert list(m2.state_dict().keys()) == keys == list(m3.state_dict().keys())
    m.double(); m.train(); m.eval()
    assert m.embracenet.docking_0.weight.dtype == torch.float64


def test_early_stopping_and_weights(ea):
    es = ea.EarlyStopping(patience=2, trace_func=lambda s: None)
    for s in (0.5, 0.6, 0.55, 0.58):
        es(s)
    assert es.early_stop and es.best_score == 0.6
    g = Golden("G5_weighted_ce")
    for case in g.meta["cases"]:
        y = torch.tensor([1] * case["pos"] + [0] * (case["B"] - case["pos"])).view(-1, 1)
        w_pos, w_neg = ea.get_loss_weights_from_labels(y)
        assert abs(w_pos - case["w_pos"]) < 1e-15 and abs(w_neg - case["w_neg"]) < 1e-15


def test_shard_rows_partition():
    from embracenet_amd import dist
    for B in (1, 7, 64, 100, 8192):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_rows(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == B
            for (a, n), (b, _) in zip(spans, spans[1:]):
                assert a + n == b


def test_fit_validates_names(ea):
    with pytest.raises(ValueError):
        ea.fit_multimodal(None, {}, {}, "cpu", "NOPE", "active_E_vs_inactive_E")
    with pytest.raises(ValueError):
        ea.fit_multimodal(None, {}, {}, "cpu", "A549", "nope")


def test_step_table_roundtrip(ea):
    t = ea.metrics.StepTable(4, "cpu")
    for k in range(3):
        ls, cs = t.slot()
        ls.fill_(0.5 * k); cs.copy_(torch.tensor([k, k + 1, k + 2, 10]))
    losses, counts = t.fetch()
    assert losses.tolist() == [0.0, 0.5, 1.0] and counts[2].tolist() == [2, 3, 4, 10] and t.n == 0


def test_inference_twin_helpers_follow_the_reference():
    """utils.py:360-375 / :178-202 restated in inference.py (host logic of row f3)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, root)
    import embracenet_amd as ea
    from oracle.configs import CONFIGS
    hp, _ = CONFIGS["cfg1"]
    single = ea.inference.get_single_model_params(hp)
    assert single["FFNN"]["n_layers"] == hp["FFNN_n_layers"] and single["CNN"]["kernel_size_l0"] == hp["CNN_kernel_size_l0"]
    assert "embracement_size" not in single["CNN"] and "n_units_l0" in single["FFNN"]
    # two blocks of (conv k=15 same padding, maxpool 10/2) on 256 positions: 124 -> 58 positions x 32 channels
    assert ea.inference.output_size_from_model_params(single["CNN"]) == 58 * hp["CNN_out_channels_l1"] == 1856
    assert ea.inference.checkpoint_name("A549", "t", 3) == "A549_EmbraceNetMultimodal_t_3_test_.pt"
    assert ea.inference.checkpoint_name("A549", "t", 3, augmentation=True) == "A549_EmbraceNetMultimodal_augmentation_t_3_test_.pt"


@pytest.mark.parametrize("native", [True, False], ids=["native-shuffle", "python-shuffle"])
def test_g11_balanced_batches_match_the_reference_sampler(native):
    """Row f4: data.BalancedBatchSampler yields the index lists of the reference's BalancePos_BatchSampler (fixture G11,
    three epochs per case: the in-place shuffles carry over from epoch to epoch), bit for bit."""
    from embracenet_amd import data
    from oracle import datagen as dg
    g = Golden("G11_balanced_batches")
    for i, case in enumerate(g.meta["cases"]):
        n, rate = case["n"], case["rate"]
        y = dg.labels(f"g11/{i}/y", n, rate).reshape(-1) if rate > 0 else np.zeros(n, dtype=np.int64)
        assert int(y.sum()) == case["positives"]
        s = data.BalancedBatchSampler(y, case["batch_size"], random_state=case["seed"], native=native)
        assert len(s) == case["len"]
        flat, sizes = [], []
        for _ in range(3):
            batches = list(iter(s))
            sizes.append([len(b) for b in batches])
            flat += [v for b in batches for v in b]
        assert sizes == case["sizes"], i
        assert np.array_equal(np.asarray(flat, dtype=np.int64), g[f"c{i}_idx"]), i
        assert len(sizes[0]) == case["len"] + 1                  # the reference yields one batch more than len()
        assert sorted(flat[:n]) == list(range(n))                # every row exactly once per epoch


@pytest.mark.parametrize("n,bs,seed", [(23, 10, 153), (200, 64, 153), (64, 64, 7), (5, 8, 1)])
def test_shuffled_batches_match_a_torch_dataloader(n, bs, seed):
    """The reference's test loader is DataLoader(..., batch_size*2, shuffle=True, generator=manual_seed(random_state+30))
    (dataprepare.py:592-594): data.ShuffledBatchSampler must walk through the same index batches, epoch after epoch."""
    from embracenet_amd import data
    ref = torch.utils.data.DataLoader(torch.arange(n), batch_size=bs, shuffle=True,
                                      generator=torch.Generator("cpu").manual_seed(seed))
    mine = data.ShuffledBatchSampler(n, bs, random_state=seed)
    assert len(mine) == len(ref)
    for _ in range(3):
        assert [b.tolist() for b in ref] == list(iter(mine))


def test_native_shuffle_equals_python_random_on_long_lists():
    """emb_mt19937_shuffle vs random.Random(seed).shuffle on lists long enough to regenerate the twister state many times
    and to exercise the rejection loop of _randbelow, with consecutive shuffles continuing one stream."""
    import random
    from embracenet_amd import data
    for seed, sizes in ((123, (50001, 1, 0, 4097, 2)), (2 ** 40 + 5, (1000, 65536))):
        rng, sh = random.Random(seed), data._PyShuffler(seed, native=True)
        for n in sizes:
            want = list(range(n))
            rng.shuffle(want)
            got = np.arange(n, dtype=np.int64)
            sh.shuffle(got)
            assert got.tolist() == want, (seed, n)


def test_library_has_no_undefined_internal_symbols():
    """ctypes binds lazily, so a kernel host stub the compiler silently dropped (seen once: buffer builtins in the host pass
    of a kernel template) would only fail at its first launch on the GPU box.  Catch it here: nothing in namespace emb may
    be left undefined in the shared object."""
    import shutil
    import subprocess
    from embracenet_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bad = [l for l in out.splitlines() if "_ZN3emb" in l]
    assert not bad, bad[:5]


def test_rebalance_trigger_is_the_reference_ratio(ea, monkeypatch):
    """Kfold_CV_Multimodal re-balances a training split exactly when the reference does: get_imbalance(y) =
    round(n_pos / n_neg, 3) < rebalance_threshold (data_pipe/utils.py:280-306, training_models_multimodal.py:533) -- a RATIO:
    9.5 % positives (ratio 0.105) train as they are at the default threshold 0.1, a positives-majority split never triggers."""
    from embracenet_amd import data, training
    assert training.get_imbalance(np.array([1] * 2 + [0] * 19)) == 0.105
    assert training.get_imbalance(n_pos=1, n_neg=3) == 0.333
    with pytest.raises(ZeroDivisionError):
        training.get_imbalance(np.ones(4))
    seen = []
    monkeypatch.setattr(data, "device_loaders", lambda *a, **k: seen.append(k.get("balanced")) or "loaders")
    cv = training.Kfold_CV_Multimodal()
    cv.rebalance_threshold, cv.random_state, cv.device, cv.precision = 0.1, 789, "cpu", "float32"
    X1, X2 = np.zeros((210, 3)), np.zeros((210, 8), dtype=np.uint8)
    y = lambda pos: np.array([1] * pos + [0] * (210 - pos))
    assert cv.build_dataloaders_forCV(X1, X2, y(20), 16, True, False) == "loaders"      # 20 / 190 = 0.105: no re-balancing
    assert cv.build_dataloaders_forCV(X1, X2, y(190), 16, True, False) == "loaders"     # ratio 9.5: never
    with pytest.raises(NotImplementedError):
        cv.build_dataloaders_forCV(X1, X2, y(17), 16, True, False)                      # 17 / 193 = 0.088 < 0.1
    with pytest.raises(NotImplementedError):
        cv.build_dataloaders_forCV(X1, X2, y(100), 16, True, True)                      # augmentation asked for
    assert cv.build_dataloaders_forCV(X1, X2, y(17), 16, False, False) == "loaders"     # test splits: never
    calls = []
    cv.rebalance = lambda X, yy, sequence, thr: (calls.append((sequence, thr)) or (X, yy))
    assert cv.build_dataloaders_forCV(X1, X2, y(17), 16, True, False) == "loaders" and calls == [(False, 0.1), (True, 0.1)]
    assert seen == [True, True, False, True]
