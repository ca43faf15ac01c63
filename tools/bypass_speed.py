"""Times the bypass_docking selection kernels (emb_embrace_bypass_fwd / _bwd) against their algorithmic HBM bytes.
forward: 2 reads + 1 write of T and 1 code byte per element; backward: 1 read of T + 1 code byte, 2 writes of T.
usage: python tools/bypass_speed.py   (MI355X)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import embracenet_amd as ea  # noqa: E402

F = ea.functional
DEV = "cuda"


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n      # us per launch


for (B, c, T) in [(1024, 768, torch.bfloat16), (4096, 1024, torch.bfloat16), (4096, 1024, torch.float32),
                  (65536, 1024, torch.bfloat16), (65536, 1024, torch.float32)]:
    x0 = torch.randn(B, c, device=DEV).to(T)
    x1 = torch.randn(B, c, device=DEV).to(T)
    dE = torch.randn(B, c, device=DEV).to(T)
    p = torch.tensor([[0.4, 0.6]], device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    sel = F.SelectInline(p, None, False, status)
    rng = F.RngState(5, 1)
    E = torch.empty_like(x0)
    code = torch.empty(B, c, dtype=torch.uint8, device=DEV)
    d0, d1 = torch.empty_like(x0), torch.empty_like(x0)
    L, ptr, st = ea._lib.lib(), ea._lib.ptr, ea._lib.stream
    dt = ea._lib.DTYPE_CODE[T]

    def fwd():
        ea._lib.check(L.emb_embrace_bypass_fwd(ptr(x0), ptr(x1), None, ptr(sel.p), 1, None, 0, ptr(status), None, rng.seed,
                                               rng.step_val, None, 0, ptr(E), ptr(code), B, c, dt, st()), "fwd")

    def bwd():
        ea._lib.check(L.emb_embrace_bypass_bwd(ptr(dE), ptr(code), ptr(d0), ptr(d1), B, c, dt, st()), "bwd")

    es = x0.element_size()
    tf, tb = timed(fwd), timed(bwd)
    bf, bb = B * c * (3 * es + 1), B * c * (3 * es + 1)
    print(json.dumps({"B": B, "c": c, "dtype": str(T), "fwd_us": round(tf, 2), "fwd_GBps": round(bf / tf / 1e3, 1),
                      "fwd_frac_hbm": round(bf / tf / 1e3 / 8000, 3), "bwd_us": round(tb, 2),
                      "bwd_GBps": round(bb / tb / 1e3, 1), "bwd_frac_hbm": round(bb / tb / 1e3 / 8000, 3)}))
