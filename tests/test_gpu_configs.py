"""GPU: WHOLE-MODEL parity at the networks of BASELINE.json configs[3] and configs[4] (bench.py WORKLOADS cfg4 / cfg5) against
oracle/ref_step.py -- the stock-PyTorch CPU restatement of the reference model, itself pinned to the imported reference by
fixtures G2 / G3 / G9 (tests/test_oracle_golden.py).

cfg4: 4 conv blocks (CNN_pre.py:24-60), one FFNN layer, c = 1024, two post layers + head (EmbraceNetMultimodal.py:134-154),
      fp32, host-RNG replay: eval logits and one train step (loss, every parameter gradient).
cfg5: 64 -> 64 k = 11 conv blocks, c = 768, d1 = 3712, bf16, TRAIN mode with the DEVICE modality-dropout gate
      (EmbraceNetMultimodal.py:178-182 on Philox, csrc/embrace_epilogue.h) replayed by the oracle's Philox.
"""
import numpy as np
import pytest
import torch

import bench
from helpers import model_batch, model_fill
from oracle import embrace_oracle as orc
from oracle import ref_step
from oracle.configs import FixedTrial

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(ea, wl, tag, rounding):
    """(oracle model in fp64, engine model in fp64 on the CPU) holding the same parameters, rounded to `rounding` so that both
    sides start from values the engine's storage type represents exactly."""
    hp, F_in = wl["hp"], wl["F"]
    rnd = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(rounding).double().numpy()
    fill = model_fill(tag)
    rfill = lambda key, shape: rnd(fill(key, shape))
    oracle_m = ref_step.OracleEmbraceNetMultimodal(hp, F_in)
    oracle_m.set_tensors(rfill)
    model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="K562", task="active_E_vs_active_P", device=DEV,
                                    in_features_FFNN=F_in).double()
    with torch.no_grad():
        for key, t in model.state_dict().items():
            if "running_" not in key and "num_batches" not in key:
                t.copy_(torch.from_numpy(rfill(key, tuple(t.shape))))
    assert sum(p.numel() for p in model.parameters()) == sum(p.numel() for p in oracle_m.parameters())
    return oracle_m, model, rnd


def test_cfg4_network_fp32_eval_and_train_step_vs_oracle(ea):
    """Tolerances (fp32 engine vs fp64 oracle on fp32-exact parameters and inputs; index tensor bit-exact):
    eval logits 1e-5 of the logit scale -- the north-star bar holds for the whole cfg4 network, running BatchNorm statistics are
    the initial (0, 1) so the four BatchNorms are plain affine maps;
    train-mode logits 2e-4: four batch-statistics BatchNorms each divide by a standard deviation formed from fp32 sums, and
    the max-pools pick by comparing neighbouring fp32 values (a different winner changes the value by the rounding distance only);
    loss 1e-5 absolute; gradients, relative to each tensor's largest gradient: 1e-4 for everything that does not sit behind a
    max-pool (docking layers, post stack, head, the epigenomic MLP, the last block's BatchNorm affine), 3e-2 for the conv-stack
    tensors -- not an engine property: stock torch in fp32 on the CPU deviates from its own fp64 run by 8e-4 ... 1.5e-2 on exactly
    these tensors at this batch (measured in the build container with oracle/ref_step.py built in fp32), because a pooling
    window whose two largest fp32 values are one rounding apart hands its whole gradient to the other position; the logits do
    not move (2e-7).  The conv tensors are additionally held to 1e-2 in relative L2 norm."""
    from embracenet_amd import training
    wl = bench.WORKLOADS["cfg4"]
    B = 256
    oracle_m, model, rnd = _pair(ea, wl, "cfg4net", torch.float32)
    assert (oracle_m.d0, oracle_m.d1, oracle_m.c, len(oracle_m.cnn), len(oracle_m.post)) == (64, 1024, 1024, 4, 3)
    model = training.prepare_model(model, DEV, "float32").set_rng("host")
    x1, x2, y = model_batch("cfg4net/b", B, wl["F"], wl["pos"])
    x1 = rnd(x1)
    X1, X2, Y = torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(y)
    g1, g2, gy = X1.to(DEV, torch.float32), X2.to(DEV, torch.float32), Y.to(DEV)
    # ---- eval
    oracle_m.eval(); model.eval()
    torch.manual_seed(21)
    with torch.no_grad():
        want = oracle_m([X1, X2]).numpy()
    torch.manual_seed(21)
    with torch.no_grad():
        got = model([g1, g2]).double().cpu().numpy()
    assert np.array_equal(model.embracenet.modality_indices().cpu().numpy(), oracle_m.last["idx"].numpy())
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err < 1e-5, err
    # ---- one train step: both branches of the modality-dropout gate (:180) over the seeds
    branches = set()
    for seed in (0, 1, 2, 3):
        oracle_m.train(); model.train()
        oracle_m.zero_grad(); model.zero_grad()
        torch.manual_seed(seed)
        out_o = oracle_m([X1, X2], is_training=True)
        loss_o = ref_step.batch_loss(out_o, Y)
        loss_o.backward()
        branches.add(oracle_m.last["t"] is not None)
        torch.manual_seed(seed)
        out_g = model([g1, g2], is_training=True)
        assert np.array_equal(model.embracenet.modality_indices().cpu().numpy(), oracle_m.last["idx"].numpy()), seed
        loss_g = ea.functional.weighted_ce(out_g, gy)
        loss_g.backward()
        want, got = out_o.detach().numpy(), out_g.detach().double().cpu().numpy()
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err < 2e-4, (seed, err)
        assert abs(loss_g.item() - loss_o.item()) < 1e-5, (seed, loss_g.item(), loss_o.item())
        params = dict(model.named_parameters())
        worst = {}
        for key in oracle_m.names:
            if "running_" in key:
                continue
            go = oracle_m.tensor(key).grad.numpy()
            gg = params[key].grad.double().cpu().numpy()
            if key.endswith(".bias") and ".CNN_model." in key and int(key.split(".")[2]) % 5 == 0:
                assert np.abs(gg).max() < 1e-5 * max(1e-3, params[key.replace(".bias", ".weight")].grad.abs().max().item())
                continue                # conv bias in front of a batch-statistics BatchNorm: exactly zero gradient, noise on both sides
            behind_pool = ".CNN_model." in key and not key.startswith(f"CNN.CNN_model.{5 * (len(oracle_m.cnn) - 1) + 1}.")
            worst[key] = (np.abs(gg - go).max() / max(np.abs(go).max(), 1e-12), 3e-2 if behind_pool else 1e-4)
            if behind_pool:
                assert np.linalg.norm(gg - go) / max(np.linalg.norm(go), 1e-12) < 1e-2, (seed, key)
        bad = {k: v for k, v in worst.items() if not v[0] < v[1]}
        assert not bad, (seed, bad)
    assert branches == {True, False}, "seeds no longer cover both modality-dropout branches"


def test_cfg5_network_bf16_train_mode_with_the_device_dropout_gate_vs_oracle(ea):
    """bf16 storage / fp32 accumulation, model.train(), rng_mode "philox": the gate, the per-row modality draw and the
    selection uniforms come from Philox on the device (kinds 1, 2, 0 of include/embrace_hip.h).  The oracle is fed the same
    draws from its own Philox restatement (pinned by the Random123 known-answer vectors) and bf16-rounded parameters / inputs.
    Index tensor bit-exact at every step, both gate branches covered; logits within 3e-2 of the logit scale (the bf16 bar of
    the cfg2 test: intermediate activations are rounded to bf16 -- FFNN layers, two pooled conv blocks, the fused output)."""
    from embracenet_amd import training
    wl = bench.WORKLOADS["cfg5"]
    B, seed, row0 = wl["B"], 4242, 2048
    oracle_m, model, rnd = _pair(ea, wl, "cfg5net", torch.bfloat16)
    assert (oracle_m.d0, oracle_m.d1, oracle_m.c) == (32, 3712, 768)
    model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=seed, row0=row0)
    c = oracle_m.c
    x1, x2, _ = model_batch("cfg5net/b", B, wl["F"], wl["pos"])
    x1 = rnd(x1)
    X1, X2 = torch.from_numpy(x1), torch.from_numpy(x2)
    g1, g2 = X1.to(DEV, torch.bfloat16), X2.to(DEV, torch.bfloat16)
    oracle_m.train(); model.train()
    rows = np.arange(row0, row0 + B, dtype=np.uint64)
    elem = rows[:, None] * np.uint64(c) + np.arange(c, dtype=np.uint64)[None, :]
    seen = set()
    for step in range(8):
        gate = orc.philox_uniform24(seed, (step << 8) | 1, np.zeros(1, np.uint64))[0]
        t = None
        if gate >= 0.5:                                     # :180-182 -- round(rand(B)): > 0.5 (an exact 0.5 rounds to even = 0)
            t = (orc.philox_uniform24(seed, (step << 8) | 2, rows) > 0.5).astype(np.int64)
        u = orc.philox_select_uniform(seed, (step << 8) | 0, elem)
        with torch.no_grad():
            want = oracle_m([X1, X2], is_training=True, inject=dict(t=t, u=u)).numpy()
            got = model([g1, g2], is_training=True)
        idx = model.embracenet.modality_indices().cpu().numpy()
        assert np.array_equal(idx, oracle_m.last["idx"].numpy()), step
        if t is not None:
            assert np.array_equal(idx, np.repeat(t[:, None], c, 1)), step       # every row used exactly one modality
        got = got.double().cpu().numpy()
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err < 3e-2, (step, err)
        assert (np.argmax(got, 1) == np.argmax(want, 1)).mean() > 0.97, step
        seen.add(t is not None)
    assert seen == {True, False}, "steps no longer cover both branches of the device gate"
