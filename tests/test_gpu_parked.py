"""GPU: the launch descriptors a deferring training step parks in the library (queued slab reductions, the rider launch of the
epigenomic MLP, the first conv block's finish / totals jobs: csrc/reduce.hip, rider.h, first_fin.h) are PER-STREAM state --
SURVEY 8b "keeps no global state ... re-entrant across streams".  Covered here: edge cases of the parked MLP backward, two
trainers interleaved on two streams, recovery after an exception (emb_reset / emb_reset_stream)."""
import numpy as np
import pytest
import torch

from helpers import model_batch
from test_gpu_model import build

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _state(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("case", ["frozen_cnn", "input_grad"])
def test_parked_mlp_backward_edge_cases(ea, case):
    """bf16, slab reductions deferred => the epigenomic MLP's backward parks itself for the conv stack's BatchNorm-backward pass.
    frozen_cnn: no conv backward follows (its parameters do not require grad) -- the optimizer launch / the flush must still find
    the parked launch (it was parked on the autograd thread; the slot belongs to the stream, not to a thread).
    input_grad: the features require grad -- autograd hands dx on as soon as the node returns, so that backward must not park.
    Both: parameters (and x.grad) bit-identical to plain launches (ride_prenets = False)."""
    from embracenet_amd import optim, training
    F = ea.functional

    def run(ride):
        model, trial, hp, F_in = build(ea, "cfg1", "pk", torch.float32)
        model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=5)
        model.ride_prenets = ride
        if case == "frozen_cnn":
            for p in model.CNN.parameters():
                p.requires_grad_(False)
        opt = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-3)
        model.train()
        xg = []
        for k in range(3):
            a, b, y = model_batch(f"pk/{k}", 96, F_in, 0.3)
            x1 = torch.from_numpy(a).to(DEV, torch.float32).requires_grad_(case == "input_grad")
            x2, yy = torch.from_numpy(b).to(DEV, torch.bfloat16), torch.from_numpy(y).to(DEV)
            opt.zero_grad()
            loss = F.weighted_ce(model([x1, x2], is_training=True), yy)
            F.reduce_defer(True)
            try:
                loss.backward()
            finally:
                F.reduce_defer(False)
            opt.step()                      # consumes the queued slabs (launching a still-parked rider first)
            F.reduce_flush()
            assert F.parked_count(all_streams=True) == 0
            if case == "input_grad":
                xg.append(x1.grad.detach().cpu().clone())
        torch.cuda.synchronize()
        return _state(model), xg
    (sa, xa), (sb, xb) = run(False), run(True)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for a, b in zip(xa, xb):
        assert torch.isfinite(a).all() and a.abs().max() > 0 and torch.equal(a, b)
    name = "FFNN.model.0.weight"
    fresh, _, _, _ = build(ea, "cfg1", "pk", torch.float32)
    assert not torch.equal(sa[name], fresh.state_dict()[name].cpu())          # the MLP did get its gradients


def test_two_trainers_interleaved_on_two_streams(ea):
    """Trainer A (stream sA) runs forward + backward of a deferring step, then trainer B (stream sB) runs whole steps, then A's
    optimizer launch consumes A's parked jobs: with per-stream lots B neither flushes nor claims nor disarms what A parked.
    Both end bit-identical to running alone."""
    from embracenet_amd import optim, training
    F = ea.functional

    def make(tag, seed):
        model, trial, hp, F_in = build(ea, "cfg1", tag, torch.float32)
        model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=seed)
        opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
        model.train()
        a, b, y = model_batch(f"{tag}/b", 64, F_in, 0.3)
        bt = (torch.from_numpy(a).to(DEV, torch.bfloat16), torch.from_numpy(b).to(DEV, torch.bfloat16), torch.from_numpy(y).to(DEV))
        return model, opt, bt

    def fwd_bwd(model, bt):
        model.zero_grad()
        loss = F.weighted_ce(model([bt[0], bt[1]], is_training=True), bt[2])
        F.reduce_defer(True)
        loss.backward()

    def finish(opt):
        F.reduce_defer(False)
        opt.step()
        F.reduce_flush()

    def alone(tag, seed, steps):
        model, opt, bt = make(tag, seed)
        for _ in range(steps):
            fwd_bwd(model, bt)
            finish(opt)
        torch.cuda.synchronize()
        return _state(model)

    ref_a, ref_b = alone("twoA", 3, 2), alone("twoB", 4, 4)
    (ma, oa, ba), (mb, ob, bb) = make("twoA", 3), make("twoB", 4)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(2):
        with torch.cuda.stream(sA):
            fwd_bwd(ma, ba)
            assert F.parked_count() > 0                 # A's jobs sit in A's lot ...
        with torch.cuda.stream(sB):
            assert F.parked_count() == 0                # ... and are invisible from B's stream
            for _ in range(2):
                fwd_bwd(mb, bb)
                finish(ob)
            assert F.parked_count() == 0
        with torch.cuda.stream(sA):
            assert F.parked_count() > 0                 # B's flushes / optimizer launches took nothing of A's
            finish(oa)
            assert F.parked_count() == 0
    torch.cuda.synchronize()
    got_a, got_b = _state(ma), _state(mb)
    for k in ref_a:
        assert torch.equal(ref_a[k], got_a[k]), ("A", k)
    for k in ref_b:
        assert torch.equal(ref_b[k], got_b[k]), ("B", k)


def test_reset_after_an_exception_leaves_no_stale_job(ea):
    """A step that raises between parking and the carrier / flush: StepRunner drops what its stream has parked (the descriptors
    point at tensors of the failed step), and the next steps equal a run that never failed."""
    from embracenet_amd import optim, training
    F = ea.functional

    def run(fail_at):
        model, trial, hp, F_in = build(ea, "cfg1", "rs", torch.float32)
        model = training.prepare_model(model, DEV, "bfloat16").set_rng("philox", seed=7)
        opt = optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
        runner = training.StepRunner(model, opt, DEV)
        table = ea.metrics.StepTable(8, DEV)
        model.train()
        for k in range(3):
            a, b, y = model_batch(f"rs/{k}", 64, F_in, 0.3)
            args = (torch.from_numpy(a).float(), torch.from_numpy(b).float(), torch.from_numpy(y), table)
            if k == fail_at:
                state, step = _state(model), model.embracenet._step_dev.clone()
                boom = RuntimeError("boom")
                orig = opt.step
                opt.step = lambda: (_ for _ in ()).throw(boom)       # forward and backward have parked their jobs by now
                with pytest.raises(RuntimeError, match="boom"):
                    runner.train_step(*args)
                opt.step = orig
                assert F.parked_count(all_streams=True) == 0
                # roll the failed step back (parameters untouched; RNG / optimizer step counters, BatchNorm buffers) and redo it
                model.load_state_dict(state)
                model.embracenet._step_dev.copy_(step)
                if hasattr(opt, "step_counter"):
                    opt.step_counter(DEV).sub_(1)
                table.n -= 1
            runner.train_step(*args)
        torch.cuda.synchronize()
        return _state(model), table.fetch()[0].tolist()
    (sa, la), (sb, lb) = run(None), run(1)
    assert la == lb, (la, lb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_reset_drops_parked_descriptors_on_every_stream(ea):
    F = ea.functional
    L, ptr, st = ea._lib.lib(), ea._lib.ptr, ea._lib.stream
    x = torch.randn(1024, 32, device=DEV)                   # (large enough for the batch-split weight gradient: a slab job)
    w = torch.randn(16, 32, device=DEV, requires_grad=True)
    b = torch.zeros(16, device=DEV, requires_grad=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for s in (s1, s2):
        with torch.cuda.stream(s):
            F.reduce_defer(True)
            F.linear(x, w, b, relu=True).sum().backward()
            assert F.parked_count() == 1
    assert F.parked_count(all_streams=True) == 2
    with torch.cuda.stream(s1):
        assert F.reset() == 1 and F.parked_count() == 0
    assert F.parked_count(all_streams=True) == 1
    assert F.reset(all_streams=True) == 1 and F.parked_count(all_streams=True) == 0
    with torch.cuda.stream(s2):                              # deferral is off again after a reset: immediate reduction
        w.grad = None
        F.linear(x, w, b, relu=True).sum().backward()
        assert F.parked_count() == 0
    torch.cuda.synchronize()
    ref = (torch.relu(x @ w.detach().t() + b.detach()) > 0).float().t() @ x
    assert torch.allclose(w.grad, ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("T", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stale_premasked_handover_is_rejected(ea, T):
    """functional._PREMASKED hands the fused head's pre-masked gradients to the fusion layer's backward, keyed by dE's address.
    Addresses repeat once the allocator recycles them, so the hand-over also carries the serial number of the forward whose code
    bytes it was made from: an entry of ANOTHER forward that happens to sit under the same addresses (forced here: every lookup
    hits a poisoned entry with matching pointer, shape and dtype) must be ignored -- gradients equal to the clean run's."""
    F = ea.functional
    B, d0, d1, c = 256, 16, 192, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(DEV)
    x0, x1, w0, w1 = mk(B, d0), mk(B, d1), mk(c, d0), mk(c, d1) * 0.2
    b0, b1 = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    cdf0, _ = F.select_prep(torch.tensor([[0.4, 0.6]], device=DEV), None, B)
    dE = mk(B, c).to(T)

    class Poisoned(dict):
        def pop(self, key, default=None):
            code_ptr = self.code_ptr
            junk = torch.full((B, c), 7.0, dtype=T, device=DEV)
            return (junk, junk.clone(), code_ptr, -1)          # right pointers, right shape and dtype, wrong forward

    def run(poison):
        leaves = [t.clone().requires_grad_() for t in (x0, x1, w0, w1)]
        E, code = F.embrace(leaves[0], leaves[1], leaves[2], b0, leaves[3], b1, cdf0, rng=F.RngState(seed=9), compute_dtype=T)
        saved = F._PREMASKED
        if poison:
            F._PREMASKED = Poisoned()
            F._PREMASKED.code_ptr = code.data_ptr()
        try:
            E.backward(dE.to(E.dtype))
        finally:
            F._PREMASKED = saved
        torch.cuda.synchronize()
        return [t.grad.clone() for t in leaves]

    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


def test_parked_copy_rides_in_the_optimizer_launch_or_runs_at_the_flush(ea):
    """emb_copy_park: a few floats copied by the next fused optimizer launch of the stream (the data-parallel step's 8 bytes of
    all-reduced class counts); emb_copy_flush runs it when no optimizer launch took it; emb_reset drops it unlaunched."""
    from embracenet_amd import optim
    F = ea.functional
    src = torch.tensor([3.0, 5.0], device=DEV)
    dst = torch.zeros(2, device=DEV)
    w = torch.nn.Parameter(torch.ones(1000, device=DEV))
    opt = optim.Adam([w], lr=1e-2)
    w.grad = torch.full_like(w, 0.5)
    F.park_copy(src, dst)
    assert F.parked_count() == 1
    torch.cuda.synchronize()
    assert dst.tolist() == [0.0, 0.0]                      # nothing has run yet
    opt.step()
    torch.cuda.synchronize()
    assert dst.tolist() == [3.0, 5.0] and F.parked_count() == 0
    assert float(w.detach().max()) < 1.0                   # the update itself happened
    src.mul_(2)
    F.park_copy(src, dst)
    F.flush_copy()
    torch.cuda.synchronize()
    assert dst.tolist() == [6.0, 10.0] and F.parked_count() == 0
    src.mul_(2)
    F.park_copy(src, dst)
    assert F.reset() == 1
    opt.step()
    torch.cuda.synchronize()
    assert dst.tolist() == [6.0, 10.0]                     # dropped, not launched
