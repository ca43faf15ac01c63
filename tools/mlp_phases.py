#!/usr/bin/env python3
"""Developer tool: phase timestamps of the fused MLP kernels (library built with -DEMB_MLP_PROF)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import embracenet_amd as ea
F = ea.functional
L = ea._lib.lib()
L.emb_debug_mlp_prof.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 32)()


def run(name, B, Fin, Ns, drops, relus):
    T = torch.bfloat16
    x = torch.rand(B, Fin, device="cuda").requires_grad_(True)
    layers, K = [], Fin
    for i, (n, p, r) in enumerate(zip(Ns, drops, relus)):
        w = ((torch.rand(n, K, device="cuda") - .5) * .2).requires_grad_(True)
        b = torch.zeros(n, device="cuda", requires_grad=True)
        layers.append((w, b, r, p, i)); K = n
    rng = F.RngState(seed=3)
    for _ in range(3):
        y = F.mlp(x.to(T), layers, rng=rng, compute_dtype=T)
        y.float().sum().backward()
    torch.cuda.synchronize()
    L.emb_debug_mlp_prof(buf)
    t = list(buf)
    f = [(t[i + 1] - t[i]) * 10 for i in range(0, 5)]
    b = [(t[i + 1] - t[i]) * 10 for i in range(8, 14)]
    print(name, "fwd ns: stage %d  layers %s" % (f[0], f[1:]), " bwd ns: stage %d layers(L3..L0) %s dx %d" % (b[0], b[1:5], b[5]), flush=True)


run("ffnn", 1024, 48, [64, 32, 16], [0.2, 0.3, 0.0], [True] * 3)
run("cfg2", 1024, 48, [32, 16, 16], [0.0, 0.0, 0.0], [True] * 3)
run("head", 1024, 256, [2], [0.0], [False])
run("post2", 1024, 256, [64, 2], [0.4, 0.0], [True, False])
