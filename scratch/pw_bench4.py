"""deep short / mid 1x1-conv layers, forward + backward, graph-timed: this library's kernels (ops.pointwise_conv: tiled GEMM with
split-K, streaming weight gradient) against the library-GEMM route (ops.library_gemm_conv) and torch's conv"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import torch.nn.functional as F
from amcontrast3d_amd import ops
DEV = 'cuda:0'
shapes = [(8, 259, 256, 3008), (8, 256, 512, 3008), (8, 768, 256, 375), (8, 256, 256, 375), (8, 512, 256, 375), (8, 384, 128, 1500),
          (8, 128, 128, 1500), (8, 256, 128, 1500), (8, 192, 64, 6000), (8, 64, 64, 6000), (8, 128, 64, 6000), (8, 512, 512, 94), (8, 1024, 512, 94)]
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
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    def run(f):
        def go_():
            xr.grad = wr.grad = None; f(xr, wr).backward(go)
        return go_
    t_own = tm(run(lambda a, b_: ops.pointwise_conv(a, b_)))
    t_lib = tm(run(lambda a, b_: ops.library_gemm_conv(a, b_))) if Ci % 4 == 0 else float('nan')
    t_conv = tm(run(lambda a, b_: F.conv1d(a, b_)))
    print(f"{Ci:4d}->{Co:4d} P={P:6d}: own {t_own:7.1f} us | library GEMMs {t_lib:7.1f} | torch conv {t_conv:7.1f}", flush=True)
