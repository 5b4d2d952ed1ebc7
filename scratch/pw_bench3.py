"""small deep 1x1-conv layers: torch conv (MIOpen wgrad) vs plain library GEMMs for all three products, graph-timed"""
import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
DEV = 'cuda:0'
shapes = [(8, 259, 256, 2976), (8, 256, 512, 2976), (8, 131, 128, 12000), (8, 768, 256, 375), (8, 256, 256, 375), (8, 384, 128, 1500),
          (8, 128, 128, 1500), (8, 192, 64, 6000), (8, 64, 64, 6000)]
def tm(fn, n=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * n) * 1000
for B, Ci, Co, P in shapes:
    x = torch.randn(B, Ci, P, device=DEV); w = torch.randn(Co, Ci, 1, device=DEV); go = torch.randn(B, Co, P, device=DEV)
    w2 = w[:, :, 0]
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    def conv_fb():
        xr.grad = wr.grad = None; F.conv1d(xr, wr).backward(go)
    f1 = lambda: torch.matmul(w2, x)
    d1 = lambda: torch.matmul(w2.t(), go)
    g1 = lambda: torch.bmm(go, x.transpose(1, 2)).sum(0)
    g2 = lambda: torch.einsum('bop,bip->oi', go, x)
    g3 = lambda: torch.matmul(go.transpose(0, 1).reshape(Co, B * P), x.transpose(0, 1).reshape(Ci, B * P).t())
    t = [tm(f) for f in (conv_fb, f1, d1, g1, g2, g3)]
    print(f"{Ci:4d}->{Co:4d} P={P:6d}: conv fwd+bwd {t[0]:7.1f} | gemm fwd {t[1]:6.1f} dgrad {t[2]:6.1f} wgrad bmm+sum {t[3]:6.1f} einsum {t[4]:6.1f} flat {t[5]:6.1f}", flush=True)
    ref = torch.einsum('bop,bip->oi', go.double(), x.double())
    print("      wgrad err", float((g1().double() - ref).abs().max() / ref.abs().max()), float((g3().double() - ref).abs().max() / ref.abs().max()))
