#!/usr/bin/env python3
"""Developer tool: isolated timings of the C-ABI entry points with HIP events (not part of the product)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import embracenet_amd as ea

dev = "cuda"
F = ea.functional
L, ptr, st = ea._lib.lib(), ea._lib.ptr, ea._lib.stream


def timeit(fn, iters=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def embrace(B, d0, d1, c, T, use_u):
    P = torch.float64 if T == torch.float64 else torch.float32
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.rand(*s, device=dev, generator=g)
    x0, x1 = r(B, d0).to(T), r(B, d1).to(T)
    w0, w1 = (r(c, d0) - .5).to(T), ((r(c, d1) - .5) * .05).to(T)
    b0 = torch.zeros(c, device=dev, dtype=P); b1 = torch.zeros(c, device=dev, dtype=P)
    cdf0, _ = F.select_prep(torch.tensor([[.58, .42]], device=dev), None, B)
    u = torch.rand(B, c, device=dev, dtype=torch.float64) if use_u else None
    E = torch.empty(B, c, dtype=T, device=dev); code = torch.empty(B, c, dtype=torch.uint8, device=dev)
    dc = ea._lib.DTYPE_CODE[T]
    def fwd():
        ea._lib.check(L.emb_embrace_fwd(ptr(x0), ptr(x1), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(cdf0), ptr(u), 3, 1, None, 0,
                                        ptr(E), ptr(code), B, d0, d1, c, dc, st()), "fwd")
    return timeit(fwd)


if __name__ == "__main__":
    for (B, d0, d1, c) in [(1024, 16, 1856, 256), (1024, 16, 1856, 512), (1024, 16, 64, 256), (1024, 16, 928, 256), (4096, 16, 1856, 256)]:
        for T in (torch.bfloat16, torch.float32):
            print(B, d0, d1, c, str(T).split('.')[-1], "philox %.2f us" % embrace(B, d0, d1, c, T, False),
                  "injected-u %.2f us" % embrace(B, d0, d1, c, T, True), flush=True)
    e = lambda: None
    t0 = torch.zeros(1, device=dev, dtype=torch.int64)
    print("counter_add launch %.2f us" % timeit(lambda: L.emb_counter_add(ptr(t0), 1, st())))
