import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import embracenet_amd as ea
F = ea.functional
torch.manual_seed(0)
B, Fin = 100, 48
T = torch.bfloat16
x = torch.rand(B, Fin, device="cuda").to(T).requires_grad_()
ws = []
K = Fin
for n in (32, 16, 16):
    ws.append(((torch.rand(n, K, device="cuda") - .5).requires_grad_(), torch.zeros(n, device="cuda", requires_grad=True)))
    K = n
layers = [(w, b, True, 0.0, 8 + i) for i, (w, b) in enumerate(ws)]
for it in range(3):
    x.grad = None
    y = F.mlp(x, layers, compute_dtype=T)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    g = x.grad.float()
    bad = (g.abs() > 1e3) | ~torch.isfinite(g)
    print("iter", it, "bad dx:", bad.sum().item(), bad.nonzero()[:10].tolist(), g[bad][:5].tolist())
    for i, (w, b) in enumerate(ws):
        gw = w.grad
        print("  layer", i, "dW finite", torch.isfinite(gw).all().item(), gw.abs().max().item(), "db", b.grad.abs().max().item())
        w.grad = None; b.grad = None
