"""GPU: the hand-written sequence pre-network (conv-as-GEMM + BN/ReLU/MaxPool, csrc/convblock.hip) against the
same stack of stock PyTorch operators evaluated in fp64 on the CPU (CNN_pre.py:37-50 semantics).
Tolerances: fp64 1e-9; fp32 2e-4 relative to the tensor's scale (BatchNorm divides by the batch std, so fp32
rounding of the conv output is amplified); bf16 4e-2 (operands rounded to bf16 first so only kernel-side
rounding is measured).  argmax / dropout decisions are integer work and are covered through their effect on the
gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import datagen as dg

pytestmark = pytest.mark.gpu
DEV = "cuda"
TD = {"f64": torch.float64, "f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f64": 1e-9, "f32": 2e-4, "bf16": 4e-2}


class Blk:
    def __init__(self, name, cin, cout, k, dtype_p):
        self.conv = torch.nn.Conv1d(cin, cout, k, padding=(k - 1) // 2).to(dtype_p)
        self.bn = torch.nn.BatchNorm1d(cout).to(dtype_p)
        with torch.no_grad():
            self.conv.weight.copy_(torch.from_numpy(dg.weight(name + "/w", (cout, cin, k), cin * k)))
            self.conv.bias.copy_(torch.from_numpy(dg.weight(name + "/b", (cout,), cin * k)))
            self.bn.weight.copy_(torch.from_numpy(dg.uniform(name + "/g", (cout,), 0.5, 1.5)))
            self.bn.bias.copy_(torch.from_numpy(dg.uniform(name + "/beta", (cout,), -0.3, 0.3)))

    def params(self):
        return [self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias]


def reference(x, blocks, training):
    h = x
    for b in blocks:
        h = TF.conv1d(h, b.conv.weight, b.conv.bias, padding=b.conv.padding[0])
        h = TF.batch_norm(h, b.bn.running_mean, b.bn.running_var, b.bn.weight, b.bn.bias, training, 0.1, 1e-5)
        h = TF.max_pool1d(TF.relu(h), 10, 2)
    return h.reshape(h.shape[0], -1)


CASES = [
    ("a549", 64, [(4, 64, 15), (64, 32, 15)]),            # trial-0 stack of the Optuna DB (d1 = 1856)
    ("one", 37, [(4, 16, 5)]),
    ("deep", 16, [(4, 32, 5), (32, 32, 5), (32, 128, 11), (128, 128, 15)]),   # BASELINE cfg4 stack, L: 256..8
    ("odd", 9, [(4, 96, 11), (96, 64, 5)]),
    ("wide", 6, [(4, 64, 5), (64, 96, 11), (96, 256, 5), (256, 512, 15)]),   # widest channel choices of the search space
    # shorter windows: several whole sequences per 256-row tile of the fused first block, last tile partly filled
    ("short100", 9, [(4, 64, 15), (64, 32, 5)], 100),
    ("short37", 21, [(4, 16, 5)], 37),
    ("cfg5", 20, [(4, 64, 11), (64, 64, 11)]),            # 64 -> 64 channels: the streaming weight-gradient kernel with 128-byte dy rows
    ("short70", 33, [(4, 64, 15), (64, 64, 15), (64, 32, 3)], 70),   # L = 31 and 11: four / eleven sequences per 128-row tile
    ("dual64", 12, [(4, 64, 15), (64, 64, 3), (64, 32, 5)]),          # 64 -> 64, k = 3: weight + input gradient in one launch (<4, 2>)
]


@pytest.mark.parametrize("dt", ["f64", "f32", "bf16"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_stack_matches_torch_reference(ea, case, training, dt):
    name, B, spec = case[:3]
    T = TD[dt]
    P = torch.float64 if dt == "f64" else torch.float32
    x = dg.onehot_sequence(f"cb/{name}/x", B)
    if len(case) > 3:
        x = np.ascontiguousarray(x[:, :, :case[3]])
    rnd = lambda t: t.to(T).double() if dt == "bf16" else t.double()
    ref_blocks = [Blk(f"cb/{name}/{i}", ci, co, k, torch.float64) for i, (ci, co, k) in enumerate(spec)]
    gpu_blocks = [Blk(f"cb/{name}/{i}", ci, co, k, P) for i, (ci, co, k) in enumerate(spec)]
    if dt == "bf16":      # the kernel sees bf16-rounded conv weights: give the reference the same values
        with torch.no_grad():
            for b in ref_blocks:
                b.conv.weight.copy_(rnd(b.conv.weight))
    if not training:
        with torch.no_grad():
            for rb, gb in zip(ref_blocks, gpu_blocks):
                for bn in (rb.bn, gb.bn):
                    bn.running_mean.copy_(torch.from_numpy(dg.uniform("cb/rm", (bn.num_features,), -0.2, 0.2)))
                    bn.running_var.copy_(torch.from_numpy(dg.uniform("cb/rv", (bn.num_features,), 0.5, 2.0)))
    xr = torch.from_numpy(x)
    out_ref = reference(xr, ref_blocks, training)
    dout = torch.from_numpy(dg.uniform(f"cb/{name}/dout", tuple(out_ref.shape), -1, 1))
    if dt == "bf16":
        dout = rnd(dout)
    out_ref.backward(dout)

    for b in gpu_blocks:
        b.conv.to(DEV); b.bn.to(DEV)
    layers = [dict(conv=b.conv, bn=b.bn, drop_p=0.0, layer_id=4 + i) for i, b in enumerate(gpu_blocks)]
    out = ea.functional.conv_stack(xr.to(DEV, P), layers, training, compute_dtype=T)
    assert out.shape == out_ref.shape
    scale = max(1.0, out_ref.abs().max().item())
    err = (out.double().cpu() - out_ref.detach()).abs().max().item() / scale
    assert err < TOL[dt], ("forward", err)
    out.backward(dout.to(DEV, T))
    for rb, gb in zip(ref_blocks, gpu_blocks):
        for pr, pg, nm in zip(rb.params(), gb.params(), ("w", "b", "gamma", "beta")):
            s = max(1e-3, pr.grad.abs().max().item())
            e = (pg.grad.double().cpu() - pr.grad).abs().max().item() / s
            if nm == "b" and training:
                continue      # mathematically zero behind BatchNorm: both sides hold rounding noise
            assert e < TOL[dt] * 20, (nm, e)
        if training:
            assert (gb.bn.running_mean.double().cpu() - rb.bn.running_mean).abs().max() < max(TOL[dt], 1e-6) * 3
            assert (gb.bn.running_var.double().cpu() - rb.bn.running_var).abs().max() < max(TOL[dt], 1e-6) * 3


def test_conv_stack_is_deterministic_and_dropout_scales(ea):
    F = ea.functional
    B, spec = 32, [(4, 32, 5), (32, 32, 11)]
    x = torch.from_numpy(dg.onehot_sequence("cbd/x", B)).to(DEV, torch.float32)
    blocks = [Blk(f"cbd/{i}", ci, co, k, torch.float32) for i, (ci, co, k) in enumerate(spec)]
    for b in blocks:
        b.conv.to(DEV); b.bn.to(DEV)
    mk = lambda p: [dict(conv=b.conv, bn=b.bn, drop_p=p, layer_id=4 + i) for i, b in enumerate(blocks)]
    a = F.conv_stack(x, mk(0.0), True, rng=F.RngState(5, 1))
    b_ = F.conv_stack(x, mk(0.0), True, rng=F.RngState(5, 1))
    assert torch.equal(a, b_)
    d1 = F.conv_stack(x, mk(0.4), True, rng=F.RngState(5, 2))
    d2 = F.conv_stack(x, mk(0.4), True, rng=F.RngState(5, 2))
    d3 = F.conv_stack(x, mk(0.4), True, rng=F.RngState(5, 3))
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    e = F.conv_stack(x, mk(0.4), False)       # eval: dropout off
    assert torch.isfinite(e).all()


def test_full_model_hip_prenets_match_stock_operators(ea):
    """same model, same batch: HIP pre-nets vs the stock torch operators (helpers.stock_prenets) agree in fp64."""
    from oracle.configs import CONFIGS, FixedTrial
    hp, F_in = CONFIGS["cfg1"]
    m = ea.EmbraceNetMultimodal(FixedTrial(hp), "A549", "active_E_vs_inactive_E", DEV, F_in).double().to(DEV).set_rng("host")
    x1 = torch.from_numpy(dg.features("fm/x1", 48, F_in)).to(DEV)
    x2 = torch.from_numpy(dg.onehot_sequence("fm/x2", 48)).to(DEV)
    m.train()
    outs = []
    import contextlib
    from helpers import stock_prenets
    for hip in (True, False):
        with (contextlib.nullcontext() if hip else stock_prenets(m)):
            for bn in [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm1d)]:
                bn.reset_running_stats()
            torch.manual_seed(11)
            out = m([x1, x2], is_training=True)
            m.zero_grad()
            out.square().sum().backward()
        outs.append((out.detach().clone(), m.CNN.CNN_model[0].weight.grad.clone(), m.FFNN.model[0].weight.grad.clone(),
                     m.CNN.CNN_model[1].running_var.clone()))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() < 1e-9 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("T", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_fused_optimizer_keeps_packed_conv_weights_current(ea, T):
    """bf16 or fp32 compute on fp32 masters: the packed images are written by the optimizer launch (emb_conv_pack_register),
    no pack kernel per step -- after a step they must equal a fresh pack of the updated parameters."""
    from embracenet_amd import optim
    F = ea.functional
    L, ptr, st = ea._lib.lib(), ea._lib.ptr, ea._lib.stream
    torch.manual_seed(3)
    blocks = [Blk(f"cb/opt/{i}", ci, co, k, torch.float32) for i, (ci, co, k) in enumerate([(4, 32, 11), (32, 16, 5)])]
    for b in blocks:
        b.conv.to(DEV); b.bn.to(DEV)
    layers = [dict(conv=b.conv, bn=b.bn, drop_p=0.0, layer_id=4 + i) for i, b in enumerate(blocks)]
    params = [p for b in blocks for p in (b.conv.weight, b.conv.bias, b.bn.weight, b.bn.bias)]
    opt = optim.Adam(params, lr=1e-2)
    x = torch.from_numpy(dg.onehot_sequence("cb/opt/x", 8)).to(DEV)
    for _ in range(3):
        opt.zero_grad()
        y = F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=T)
        y.float().square().mean().backward()
        opt.step()
    torch.cuda.synchronize()
    cin_pad = 8 if T == torch.bfloat16 else 4
    bits = torch.int16 if T == torch.bfloat16 else torch.int32
    for i, b in enumerate(blocks):
        w = b.conv.weight
        Cout, Cin, k = w.shape
        cached = F._PACKS[id(w)]
        wpack, wflip = F.conv_packed(w, T, cin_pad, need_flip=i > 0)      # cache hit: optimizer-maintained buffers
        assert wpack.data_ptr() == cached[2].data_ptr() and F._PACKS[id(w)] is cached, f"block {i}: re-packed, not maintained"
        ref_pack = torch.empty_like(wpack)
        ref_flip = torch.empty_like(wflip) if wflip is not None else None
        ea._lib.check(L.emb_conv_pack_weight(ptr(w.detach()), ptr(ref_pack), ptr(ref_flip), Cout, Cin, cin_pad, k,
                                             ea._lib.DTYPE_CODE[T], st()), "pack")
        torch.cuda.synchronize()
        assert torch.equal(wpack.view(bits), ref_pack.view(bits)), f"block {i}: wpack stale"
        if wflip is not None:
            assert torch.equal(wflip.view(bits), ref_flip.view(bits)), f"block {i}: wflip stale"
        cin_pad = Cout


@pytest.mark.parametrize("spec", [[(4, 64, 15), (64, 32, 15)], [(4, 16, 5)], [(4, 96, 11), (96, 64, 5)]],
                         ids=["fused2", "fused1", "unfused"])
def test_base_code_input_equals_onehot_input(ea, spec):
    """Row f4: staging the DNA window as one byte per position (functional.pack_onehot) gives bit-identical outputs and
    parameter gradients to the [B, 4, L] one-hot tensor (bf16 path; 'unfused' = a first block the fused kernels do not take)."""
    F = ea.functional
    B = 21
    x = dg.onehot_sequence("cb/codes/x", B)
    x[:, :, 5] = 0                                         # an all-zero column (unknown base)
    xt = torch.from_numpy(x).to(DEV)
    codes = F.pack_onehot(xt)
    assert codes.dtype == torch.uint8 and codes.shape == (B, 256) and int(codes[0, 5]) == 4
    assert torch.equal(torch.nn.functional.one_hot(codes.long(), 5)[..., :4].permute(0, 2, 1).float(), xt.float())
    results = []
    for inp in (xt.float(), codes):
        blocks = [Blk(f"cb/codes/{i}", ci, co, k, torch.float32) for i, (ci, co, k) in enumerate(spec)]
        for b in blocks:
            b.conv.to(DEV); b.bn.to(DEV)
        layers = [dict(conv=b.conv, bn=b.bn, drop_p=0.0, layer_id=4 + i) for i, b in enumerate(blocks)]
        out = F.conv_stack(inp, layers, True, rng=F.RngState(seed=5), compute_dtype=torch.bfloat16)
        out.backward(torch.ones_like(out))
        results.append((out.detach(), [p.grad.clone() for b in blocks for p in b.params()]))
    (o1, g1), (o2, g2) = results
    assert torch.equal(o1, o2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", ["f64", "bf16"])
@pytest.mark.parametrize("name,B,split,spec", [("a549", 24, 16, [(4, 64, 15), (64, 32, 15)]),
                                                ("deep", 10, 3, [(4, 32, 5), (32, 32, 5), (32, 128, 11)])])
def test_global_batch_batchnorm_over_two_shards_equals_one_process(ea, name, B, split, spec, dt):
    """with bn_sync the sharded stack reproduces the single-process stack -- outputs, BatchNorm running
    statistics, and parameter gradients summed over the shards.  Two 'ranks' run one after the other on the one GPU; the
    all-reduce callable is emulated by fixed-point passes: pass p replays the global sums of exchange points < p (known
    from the pass before) and records the local sums of point p, so after 2*n_layers+1 passes every exchange used the
    true global value -- exactly what an all-reduce delivers.  Dropout is on: its mask is keyed on the global row."""
    F = ea.functional
    T = TD[dt]
    P = torch.float64 if dt == "f64" else torch.float32
    x = torch.from_numpy(dg.onehot_sequence(f"sbn/{name}/x", B)).to(DEV, P)
    def blocks():
        bl = [Blk(f"sbn/{name}/{i}", ci, co, k, P) for i, (ci, co, k) in enumerate(spec)]
        for b in bl:
            b.conv.to(DEV); b.bn.to(DEV)
        return bl, [dict(conv=b.conv, bn=b.bn, drop_p=0.3 if i == 0 else 0.0, layer_id=4 + i) for i, b in enumerate(bl)]
    full, layers = blocks()
    out_full = F.conv_stack(x, layers, True, rng=F.RngState(7, 3, None, 0), compute_dtype=T)
    dout = torch.from_numpy(dg.uniform(f"sbn/{name}/dout", tuple(out_full.shape), -1, 1)).to(DEV, T)
    out_full.backward(dout)

    shards = [(0, split), (split, B - split)]
    n_sync = 2 * len(spec)
    known = []                                                 # global sums of exchange points 0 .. p-1
    for p in range(n_sync + 1):
        local, outs, ranks = [], [], []
        for r, (row0, rows) in enumerate(shards):
            bl, ly = blocks()
            seen = []
            def sync(t, seen=seen):
                i = len(seen)
                seen.append(t.clone())
                if i < len(known):
                    t.copy_(known[i])
                return t
            o = F.conv_stack(x[row0:row0 + rows], ly, True, rng=F.RngState(7, 3, None, row0), compute_dtype=T, bn_sync=sync)
            o.backward(dout[row0:row0 + rows])
            assert len(seen) == n_sync
            local.append(seen); outs.append(o); ranks.append(bl)
        if p < n_sync:
            known.append(local[0][p] + local[1][p])
    assert float(known[0][-1]) == B * 256                      # the row count rides along in the exchanged vector

    tol = 1e-9 if dt == "f64" else 3e-2
    got = torch.cat(outs).double()
    assert (got - out_full.double()).abs().max().item() <= tol * max(1.0, out_full.double().abs().max().item())
    for i, fb in enumerate(full):
        for j, nm in enumerate(("w", "b", "gamma", "beta")):
            ref = fb.params()[j].grad.double()
            summed = ranks[0][i].params()[j].grad.double() + ranks[1][i].params()[j].grad.double()
            if nm == "b":
                continue                                       # mathematically zero behind BatchNorm
            assert (summed - ref).abs().max().item() <= tol * 20 * max(1e-3, ref.abs().max().item()), (i, nm)
        for rk in ranks:                                       # every rank tracks the global statistics
            assert (rk[i].bn.running_mean.double() - fb.bn.running_mean.double()).abs().max() < max(tol, 1e-6)
            assert (rk[i].bn.running_var.double() - fb.bn.running_var.double()).abs().max() < max(tol, 1e-6)
            assert int(rk[i].bn.num_batches_tracked) == 1


@pytest.mark.parametrize("spec,L", [([(4, 64, 15), (64, 32, 15)], 256), ([(4, 16, 5)], 256), ([(4, 32, 11), (32, 32, 5)], 100)],
                         ids=["a549", "one", "short100"])
def test_loader_layout_staged_by_the_statistics_pass_equals_the_conversion_launch(ea, spec, L):
    """Training step on [B, 4, L] windows already in the compute dtype: the first block's statistics pass reads the loader's
    layout itself and writes the channels-last image (x_codes = 2) -- outputs, gradients and BatchNorm statistics must be
    bit-identical to the path through emb_ncl_to_nlc (taken here by handing the same values over in fp32)."""
    F = ea.functional
    B = 21
    x = torch.from_numpy(np.ascontiguousarray(dg.onehot_sequence("ncl2/x", B)[:, :, :L])).to(DEV)
    res = []
    for dtype_in in (torch.bfloat16, torch.float32):
        blocks = [Blk(f"ncl2/{i}", ci, co, k, torch.float32) for i, (ci, co, k) in enumerate(spec)]
        for b in blocks:
            b.conv.to(DEV); b.bn.to(DEV)
        layers = [dict(conv=b.conv, bn=b.bn, drop_p=0.2 if i == 0 else 0.0, layer_id=4 + i) for i, b in enumerate(blocks)]
        out = F.conv_stack(x.to(dtype_in), layers, True, rng=F.RngState(seed=5, step_val=2), compute_dtype=torch.bfloat16)
        out.backward(torch.ones_like(out) * 0.01)
        res.append((out.detach(), [p.grad.clone() for b in blocks for p in b.params()],
                    [b.bn.running_var.clone() for b in blocks]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,L,spec", [(40, 256, [(4, 64, 15), (64, 32, 15)]), (21, 37, [(4, 16, 5)]), (9, 100, [(4, 32, 11)]),
                                      (33, 70, [(4, 64, 7), (64, 32, 3)])], ids=["a549", "short37", "one100", "short70"])
def test_first_block_linear_backward_equals_the_recomputing_backward(ea, B, L, spec):
    """bf16 first block: the recompute-free backward (csrc/first_gram.h: dW, dgamma, dbeta from A = g^T xview, the lag statistics
    of the input and the weights) against the two recomputing passes it replaces, on a REAL-valued input (the lag statistics must
    not rely on one-hot rows) and against autograd in fp64.  Both GPU paths see the same bf16 inputs and round the dense gradient
    tile to bf16 once, in different places: they agree with each other far more closely (measured 2e-3 .. 2e-2 of the largest
    gradient; bar 2.5e-2) than either agrees with fp64 arithmetic on bf16 activations -- an error of a few percent in the
    lag-statistics path would show up as a disagreement of the two paths."""
    L_ = ea._lib.lib()
    x = torch.from_numpy(dg.uniform(f"lin/{B}/{L}/x", (B, 4, L), -1.0, 1.0)).to(torch.bfloat16).double()
    ref_blocks = [Blk(f"lin/{B}/{i}", ci, co, k, torch.float64) for i, (ci, co, k) in enumerate(spec)]
    with torch.no_grad():
        for b in ref_blocks:
            b.conv.weight.copy_(b.conv.weight.to(torch.bfloat16).double())
    out_ref = reference(x, ref_blocks, True)
    dout = torch.from_numpy(dg.uniform(f"lin/{B}/dout", tuple(out_ref.shape), -1, 1)).to(torch.bfloat16).double()
    out_ref.backward(dout)

    def run(linear):
        was = L_.emb_convblock_first_linear(int(linear))
        try:
            blocks = [Blk(f"lin/{B}/{i}", ci, co, k, torch.float32) for i, (ci, co, k) in enumerate(spec)]
            for b in blocks:
                b.conv.to(DEV); b.bn.to(DEV)
            layers = [dict(conv=b.conv, bn=b.bn, drop_p=0.0, layer_id=4 + i) for i, b in enumerate(blocks)]
            out = ea.functional.conv_stack(x.to(DEV, torch.float32), layers, True, compute_dtype=torch.bfloat16)
            out.backward(dout.to(DEV, torch.bfloat16))
            torch.cuda.synchronize()
            return out.double().cpu(), [p.grad.double().cpu() for p in blocks[0].params()]
        finally:
            L_.emb_convblock_first_linear(was)
    out_l, g_l = run(True)
    out_r, g_r = run(False)
    assert torch.equal(out_l, out_r)                       # the forward is the same computation in both modes
    ref = [p.grad for p in ref_blocks[0].params()]
    for nm, a, b, r in zip(("w", "b", "gamma", "beta"), g_l, g_r, ref):
        if nm == "b":
            assert a.abs().max().item() == 0.0               # exactly zero behind training-mode BatchNorm
            continue
        s = max(1e-3, r.abs().max().item())
        e_lin, e_rec = (a - r).abs().max().item() / s, (b - r).abs().max().item() / s
        # measured on MI355X (largest error relative to the largest gradient): the two paths agree to 1.7e-3 .. 1.6e-2; against fp64
        # arithmetic both carry the SAME error of the bf16 activations / dense gradient tile (weights 7e-2 .. 1.7e-1, relative L2
        # 8e-2 .. 1.2e-1; gamma / beta of a single-block stack 2e-3 .. 4e-3, behind a second block 6e-2 .. 9e-2)
        assert e_lin < 0.25 and e_rec < 0.25, (nm, e_lin, e_rec)
        assert ((a - r).norm() / r.norm()).item() < 0.15 and ((b - r).norm() / r.norm()).item() < 0.15, nm
        if len(spec) == 1 and nm in ("gamma", "beta"):
            assert e_lin < 1e-2 and e_rec < 1e-2, (nm, e_lin, e_rec)
        assert (a - b).abs().max().item() / s < 2.5e-2, (nm, "paths disagree", (a - b).abs().max().item() / s)
