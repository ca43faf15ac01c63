#!/usr/bin/env python3
"""Developer tool: phase timestamps of the direct conv kernels (library built with -DEMB_CONV_PROF)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import embracenet_amd as ea
F = ea.functional
L = ea._lib.lib()
PROF = hasattr(L, "emb_debug_conv_prof")
if PROF:
    L.emb_debug_conv_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()


def stack(chans, ks):
    layers = []
    for i, (ci, co, k) in enumerate(zip(chans[:-1], chans[1:], ks)):
        conv = nn.Conv1d(ci, co, k, padding=(k - 1) // 2).cuda()
        bn = nn.BatchNorm1d(co).cuda()
        layers.append(dict(conv=conv, bn=bn, drop_p=0.0, layer_id=4 + i))
    return layers


def run(name, B, chans, ks, Lseq, fam):
    layers = stack(chans, ks)
    x = torch.rand(B, chans[0], Lseq, device="cuda")
    if PROF:
        L.emb_debug_conv_prof(buf, fam)
    for _ in range(3):
        y = F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=torch.bfloat16)
        y.float().sum().backward()
    torch.cuda.synchronize()
    if not PROF:
        return
    L.emb_debug_conv_prof(buf, fam)
    t = [v for v in buf]
    t0 = t[0]
    print(name, "fam", fam, " ".join("%d:%d" % (i, (v - t0) * 10) for i, v in enumerate(t) if v), flush=True)


for fam in (0, 1):
    run("L0   4->64 k15 L256", 1024, [4, 64], [15], 256, fam)
    run("L1  64->32 k15 L124", 1024, [64, 32], [15], 124, fam)
    run("L0+L1 (last of family)", 1024, [4, 64, 32], [15, 15], 256, fam)
