#!/usr/bin/env python3
"""Developer tool: time the direct conv kernels with phases switched off (library built with -DEMB_CONV_PROF)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import embracenet_amd as ea
L, ptr, st = ea._lib.lib(), ea._lib.ptr, ea._lib.stream
T = torch.bfloat16
DC = ea._lib.DTYPE_CODE[T]


def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def layer(B, Lq, cin, cout, k):
    """forward + backward of one conv block through the C ABI; returns closures"""
    x = torch.rand(B, Lq, cin, device="cuda").to(T)
    w = (torch.rand(cout, cin, k, device="cuda") - .5)
    bias = torch.zeros(cout, device="cuda")
    KK = k * cin
    wp = torch.empty(cout, KK, device="cuda", dtype=T); wf = torch.empty(cin, k * cout, device="cuda", dtype=T)
    ea._lib.check(L.emb_conv_pack_weight(ptr(w), ptr(wp), ptr(wf), cout, cin, cin, k, DC, st()), "pack")
    return x, w, bias, wp, wf


if __name__ == "__main__":
    import inspect
    F = ea.functional
    import torch.nn as nn
    names = {0: "k-loop", 1: "bias", 2: "prefetch issue", 3: "LDS commit", 4: "out stores", 5: "W staging"}
    for (desc, chans, Lq) in (("L0 4->64 L256", [4, 64], 256), ("L1 64->32 L124", [64, 32], 124)):
        conv = nn.Conv1d(chans[0], chans[1], 15, padding=7).cuda(); bn = nn.BatchNorm1d(chans[1]).cuda()
        layers = [dict(conv=conv, bn=bn, drop_p=0.0, layer_id=4)]
        x = torch.rand(1024, chans[0], Lq, device="cuda")
        for bits in (0, 1, 2, 4, 8, 16, 32, 1 | 4 | 8, 63):
            L.emb_debug_conv_dbg(bits)
            y = F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=T)
            g = torch.ones_like(y)
            tf = timeit(lambda: F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=T), 30, 3)
            def fb():
                yy = F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=T)
                yy.backward(g)
            tb = timeit(fb, 30, 3)
            off = "+".join(names[b] for b in range(6) if bits >> b & 1) or "none"
            print("%-16s off[%-40s] fwd-only %.1f us   fwd+bwd %.1f us" % (desc, off, tf, tb), flush=True)
        L.emb_debug_conv_dbg(0)
