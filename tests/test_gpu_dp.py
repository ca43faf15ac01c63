"""GPU, two processes on the one device (gloo): the data-parallel train step on the REAL kernels reproduces the
single-process step on the global batch: rows sharded by training.StepRunner, class weights from the global
counts, gradients summed, modality-dropout gate and selection uniforms keyed on the global row, BatchNorm over the global
batch (dist.set_sync_batchnorm).  fp64: everything but the summation order is identical -> 1e-9; fp32 (loss inside the
classifier-head launch) -> 1e-4."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, B = 3, 64


def _run(world, rank, port, precision, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import embracenet_amd as ea
    from embracenet_amd import dist, optim, training
    from helpers import model_batch
    from oracle.configs import CONFIGS, FixedTrial
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    hp, F_in = CONFIGS["small"]
    torch.manual_seed(7)
    model = ea.EmbraceNetMultimodal(FixedTrial(hp), cell_line="A549", task="active_E_vs_inactive_E", device=dev,
                                    in_features_FFNN=F_in)
    model = training.prepare_model(model, dev, precision).set_rng("philox", seed=13)
    dist.set_sync_batchnorm(model)
    opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
    runner = training.StepRunner(model, opt, dev)
    table = ea.metrics.StepTable(STEPS + 3, dev)
    model.train()
    cast = torch.float64 if precision == "float64" else torch.float32
    for k in range(STEPS):
        a, b, y = model_batch(f"dpgpu/{k}", B, F_in, 0.3)
        runner.train_step(torch.from_numpy(a).to(cast), torch.from_numpy(b).to(cast), torch.from_numpy(y), table)   # the GLOBAL batch
    np.save(out + f".idx{rank}.npy", model.embracenet.modality_indices().cpu().numpy())      # selection of the last sharded step, local rows
    # a global batch with fewer rows than ranks (ragged last batch): every rank runs it whole, rank 0 alone contributes
    a, b, y = model_batch("dpgpu/tiny", 1, F_in, 0.3)
    runner.train_step(torch.from_numpy(a).to(cast), torch.from_numpy(b).to(cast), torch.from_numpy(y), table)
    runner.eval_step(torch.from_numpy(a).to(cast), torch.from_numpy(b).to(cast), torch.from_numpy(y), table)
    losses, counts = table.fetch()
    if world > 1:                                                # per-shard loss shares / counts -> global
        t = torch.from_numpy(np.concatenate([losses, counts.reshape(-1).astype(np.float64)]))
        torch.distributed.all_reduce(t)
        losses, counts = t[:STEPS + 2].numpy(), t[STEPS + 2:].numpy().reshape(-1, 4)
    if rank == 0:
        np.savez(out, losses=losses, counts=counts,
                 **{k.replace(".", "__"): v.detach().double().cpu().numpy() for k, v in model.state_dict().items()})
    if world > 1:
        assert runner.flat is not None and len(runner.flat.buckets()) == 2, "two-bucket flat gradients expected under DP"
        for p in model.parameters():                             # gradients are views of the flat buckets
            assert any(p.grad.data_ptr() >= bk.flat.data_ptr() and
                       p.grad.data_ptr() < bk.flat.data_ptr() + bk.flat.numel() * bk.flat.element_size() for bk in runner.flat.buckets())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _worker(rank, world, port, precision, out):
    _run(world, rank, port, precision, out)


@pytest.mark.parametrize("precision", ["float64", "float32"])
def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path, precision):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(1, port, precision, one), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, port + 1, precision, two), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    tol = 1e-9 if precision == "float64" else 1e-4
    assert np.abs(a["losses"] - b["losses"]).max() < max(tol, 1e-6), (a["losses"], b["losses"])   # (the table stores fp32)
    assert np.array_equal(a["counts"][:, 2:], b["counts"][:, 2:])          # positives / rows of the global batch
    # the sampled modality index of every (row, feature) does not depend on the sharding: bit-exact
    idx1 = np.load(one + ".idx0.npy")
    idx2 = np.concatenate([np.load(two + f".idx{r}.npy") for r in range(2)])
    assert idx1.shape == (B, idx1.shape[1]) and np.array_equal(idx1, idx2)
    for k in a.files:
        if k in ("losses", "counts"):
            continue
        err = np.abs(a[k] - b[k]).max()
        assert err <= tol * max(1.0, np.abs(a[k]).max()), (k, err)
