"""1x1-convolution layers of PointNeXt-S at B=8, N=24000: this package's kernels vs torch (MIOpen / rocBLAS), forward
and forward+backward, timed as hipGraph replays (no launch overhead, like the bench's captured step)."""
import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
from amcontrast3d_amd import ops
DEV = 'cuda:0'
shapes = [(8, 259, 256, (93, 32)), (8, 256, 512, (93, 32)), (8, 131, 128, (375, 32)), (8, 128, 256, (375, 32)),
          (8, 768, 256, (375,)), (8, 256, 256, (375,)), (8, 384, 128, (1500,)), (8, 128, 128, (1500,)),
          (8, 192, 64, (6000,)), (8, 64, 64, (6000,)), (8, 96, 32, (24000,)), (8, 32, 32, (24000,))]
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
for B, Ci, Co, sp in shapes:
    x = torch.randn(B, Ci, *sp, device=DEV, requires_grad=True); w = torch.randn(Co, Ci, *([1] * len(sp)), device=DEV, requires_grad=True)
    go = torch.randn(B, Co, *sp, device=DEV)
    conv = F.conv1d if len(sp) == 1 else F.conv2d
    P = x[0, 0].numel()
    def mine_f():
        with torch.no_grad(): ops.pointwise_conv(x, w)
    def mine_fb():
        x.grad = w.grad = None; ops.pointwise_conv(x, w).backward(go)
    def lib_f():
        with torch.no_grad(): conv(x, w)
    def lib_fb():
        x.grad = w.grad = None; conv(x, w).backward(go)
    gf = 2 * B * P * Ci * Co / 1e9
    t = [tm(f) for f in (mine_f, mine_fb, lib_f, lib_fb)]
    print(f"{Ci:4d}->{Co:4d} P={P:6d} ({gf:5.2f} GF): mine fwd {t[0]:7.1f} us ({gf/t[0]*1e3:5.1f} TF) fwd+bwd {t[1]:7.1f} ({3*gf/t[1]*1e3:5.1f} TF) | torch fwd {t[2]:7.1f} fwd+bwd {t[3]:7.1f}", flush=True)
