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
if len(sys.argv) > 1 and sys.argv[1] == "xl":  # the pointwise convs of PointNeXt-XL's InvResMLP blocks at 2 x 64000 / 1 x 120000 points
    shapes = [(2, 64, 256, 16000), (2, 256, 64, 16000), (2, 128, 512, 4000), (2, 512, 128, 4000), (2, 256, 1024, 1000),
              (2, 1024, 256, 1000), (2, 512, 2048, 250), (2, 2048, 512, 250), (1, 64, 256, 30000), (1, 128, 512, 7500),
              (1, 512, 128, 7500), (1, 256, 1024, 1875), (1, 512, 2048, 468), (2, 64, 64, 16000), (2, 128, 128, 4000)]
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
