#!/usr/bin/env python3
"""Developer tool: phase timestamps of the fused first conv block (library built with -DEMB_CONV_PROF)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import embracenet_amd as ea
F = ea.functional
L = ea._lib.lib()
L.emb_debug_first_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
conv = nn.Conv1d(4, 64, 15, padding=7).cuda(); bn = nn.BatchNorm1d(64).cuda()
conv2 = nn.Conv1d(64, 32, 15, padding=7).cuda(); bn2 = nn.BatchNorm1d(32).cuda()
layers = [dict(conv=conv, bn=bn, drop_p=0.0, layer_id=4), dict(conv=conv2, bn=bn2, drop_p=0.0, layer_id=5)]
x = torch.rand(1024, 4, 256, device="cuda")
for mode in (0, 1, 4):
    L.emb_debug_first_prof(buf, mode)
    for _ in range(5):
        y = F.conv_stack(x, layers, True, rng=F.RngState(seed=1), compute_dtype=torch.bfloat16)
        y.float().sum().backward()
    torch.cuda.synchronize()
    L.emb_debug_first_prof(buf, mode)
    t = [v for v in buf]
    print("mode", mode, " ".join("%d:%d" % (i, (v - t[0]) * 10) for i, v in enumerate(t) if v), flush=True)
