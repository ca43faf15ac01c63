"""GPU: classifier head + weighted CE + head backward in one launch (csrc/head.hip) against the oracle (numpy, fp64) and
against the three-launch path it replaces.  fp32 tolerance 1e-5 relative (north-star tolerance for fp32 logits); bf16
inputs are rounded first so that only kernel-side rounding is measured; confusion counts are integers -> exact."""
import numpy as np
import pytest
import torch

from oracle import datagen as dg
from oracle import embrace_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle(E, W, b, y, logits_seen):
    """loss / dlogits from the logits the kernel reports (they are rounded to the activation type), then the head backward."""
    n = len(y)
    pos = int(y.sum())
    w = orc.class_weights(y).astype(np.float64)                 # [w_neg, w_pos]
    z = logits_seen.astype(np.float64)
    zmax = z.max(1, keepdims=True)
    lse = zmax[:, 0] + np.log(np.exp(z - zmax).sum(1))
    wy = w[y]
    den = w[1] * pos + w[0] * (n - pos)
    loss = (wy * (lse - z[np.arange(n), y])).sum() / den
    oh = np.zeros_like(z); oh[np.arange(n), y] = 1
    dz = (np.exp(z - lse[:, None]) - oh) * (wy / den)[:, None]
    return loss, dz @ W, dz.T @ E, dz.sum(0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,K,rate", [(64, 256, 0.3), (1024, 256, 0.1), (37, 128, 0.5), (203, 1024, 0.2), (9, 4, 0.4), (50, 260, 0.0)])
def test_head_ce_matches_oracle(ea, B, K, rate, dt):
    F = ea.functional
    T = torch.float32 if dt == "f32" else torch.bfloat16
    E = dg.uniform(f"head/{B}/{K}/E", (B, K), -1, 1)
    W = dg.weight(f"head/{B}/{K}/W", (2, K), K) * 4
    b = dg.weight(f"head/{B}/{K}/b", (2,), K)
    y = dg.labels(f"head/{B}/{K}/y", B, rate).reshape(-1) if rate > 0 else np.zeros(B, dtype=np.int64)
    Et = torch.from_numpy(E).to(DEV, T).requires_grad_(True)
    Wt = torch.from_numpy(W).to(DEV, torch.float32).requires_grad_(True)
    bt = torch.from_numpy(b).to(DEV, torch.float32).requires_grad_(True)
    E64, W64, b64 = Et.detach().double().cpu().numpy(), Wt.detach().double().cpu().numpy(), bt.detach().double().cpu().numpy()
    counts = torch.zeros(2, dtype=torch.int64, device=DEV)
    loss = torch.zeros(1, dtype=torch.float32, device=DEV)
    conf = torch.zeros(4, dtype=torch.int64, device=DEV)
    ticks = (torch.full((1,), 5, dtype=torch.int64, device=DEV), torch.full((1,), 9, dtype=torch.int64, device=DEV))
    arm = F.FusedLoss(torch.from_numpy(y).to(DEV), counts, False, loss, conf, ticks)
    assert F.head_ce_supported(B, K, T)
    logits = F.head_ce(Et, Wt, bt, arm, compute_dtype=T)
    z_ref = E64 @ W64.T + b64
    tol = 1e-5 if dt == "f32" else 1e-2
    assert np.abs(logits.detach().double().cpu().numpy() - z_ref).max() <= tol * max(1.0, np.abs(z_ref).max())
    logits.backward(torch.zeros_like(logits))                    # the incoming gradient is ignored by contract
    seen = logits.detach().double().cpu().numpy()
    l_ref, dE_ref, dW_ref, db_ref = _oracle(E64, W64, b64, y, seen)
    assert counts.tolist() == [int(y.sum()), B]
    assert abs(float(loss) - l_ref) <= 2e-6 * max(1.0, abs(l_ref))
    pred = (seen[:, 1] > seen[:, 0]).astype(np.int64)
    assert conf.tolist() == [int((pred & y).sum()), int(pred.sum()), int(y.sum()), B]
    assert [int(t) for t in ticks] == [6, 10]
    gtol = 2e-5 if dt == "f32" else 1e-2
    for got, ref, nm in ((Et.grad, dE_ref, "dE"), (Wt.grad, dW_ref, "dW"), (bt.grad, db_ref, "db")):
        err = np.abs(got.double().cpu().numpy() - ref).max()
        assert err <= gtol * max(1e-6, np.abs(ref).max()), (nm, err)


def test_head_ce_eval_and_global_counts(ea):
    """No-grad call: loss / counts are final after the forward; given global class counts are used, not recounted."""
    F = ea.functional
    B, K = 100, 64
    E = torch.from_numpy(dg.uniform("head/ev/E", (B, K), -1, 1)).to(DEV, torch.float32)
    W = torch.from_numpy(dg.weight("head/ev/W", (2, K), K)).to(DEV, torch.float32)
    b = torch.zeros(2, device=DEV)
    y = dg.labels("head/ev/y", B, 0.3).reshape(-1)
    yt = torch.from_numpy(y).to(DEV)
    out = {}
    for name, counts, glob in (("local", torch.zeros(2, dtype=torch.int64, device=DEV), False),
                               ("global", torch.tensor([77, 300], dtype=torch.int64, device=DEV), True)):
        loss = torch.zeros(1, device=DEV); conf = torch.zeros(4, dtype=torch.int64, device=DEV)
        with torch.no_grad():
            z = F.head_ce(E, W, b, F.FusedLoss(yt, counts, glob, loss, conf))
        ref, _ = F.weighted_ce_with_grad(z, yt, class_counts=counts.clone(), global_counts=glob)
        assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref))), name
        assert conf[2:].tolist() == [int(y.sum()), B]
        out[name] = float(loss)
    assert out["local"] != out["global"]


@pytest.mark.parametrize("precision", ["float32", "bfloat16"])
def test_fused_loss_training_tracks_the_three_launch_path(ea, precision):
    """A few StepRunner steps with the loss inside the head launch vs as its own launch: same trajectory up to rounding."""
    from embracenet_amd import optim, training
    from oracle.configs import CONFIGS, FixedTrial
    from helpers import model_batch
    hp, F_in = CONFIGS["small"]
    res = {}
    for fuse in (True, False):
        model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="A549", task="active_E_vs_inactive_E", device=DEV,
                                        in_features_FFNN=F_in)
        torch.manual_seed(2)
        model.apply(ea.metrics.weight_reset)
        model = training.prepare_model(model, DEV, precision).set_rng("philox", seed=3)
        opt = optim.Adam(model.parameters(), lr=1e-3)
        runner = training.StepRunner(model, opt, DEV)
        runner.fuse_loss = fuse
        assert model.fused_loss_ready(64)
        table = ea.metrics.StepTable(16, DEV)
        model.train()
        for k in range(6):
            a, b, y = model_batch(f"fl/{k}", 64, F_in, 0.3)
            runner.train_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
        model.eval()
        a, b, y = model_batch("fl/eval", 64, F_in, 0.3)
        runner.eval_step(torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
        losses, counts = table.fetch()
        res[fuse] = (losses, counts, model.embracenet.docking_1.weight.detach().float().cpu().numpy())
    tol = 1e-4 if precision == "float32" else 3e-2
    assert np.abs(res[True][0] - res[False][0]).max() < tol * max(1.0, np.abs(res[False][0]).max())
    assert np.array_equal(res[True][1][:, 2:], res[False][1][:, 2:])            # positives / rows of every step
    assert np.abs(res[True][2] - res[False][2]).max() < tol
