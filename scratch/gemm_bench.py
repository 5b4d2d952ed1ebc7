"""fp32 vs bf16 pointwise-conv kernels on the shapes of PointNeXt-L / XL (graph-free, event-timed):
python scratch/gemm_bench.py"""
import sys, torch
sys.path.insert(0, '.')
from amcontrast3d_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
shapes = [(8, 64, 256, 6000), (8, 256, 64, 6000), (8, 128, 512, 1500), (8, 512, 128, 1500), (8, 256, 1024, 375), (8, 1024, 256, 375),
          (8, 512, 2048, 93), (1, 64, 256, 30000), (1, 256, 1024, 1875), (1, 1024, 4096, 468), (1, 4096, 1024, 468), (8, 67, 64, 24000)]
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, ci, co, P in shapes:
    x = torch.randn(B, ci, P, device=dev, requires_grad=True); w = torch.randn(co, ci, 1, device=dev, requires_grad=True)
    go = torch.randn(B, co, P, device=dev)
    fl = 2.0 * B * ci * co * P
    row = f"B{B} {ci:5d}->{co:5d} P{P:6d}: "
    for name, bf in (("f32", False), ("bf16", True)):
        f = t(lambda: ops.pointwise_conv(x, w, None, bf))
        y = ops.pointwise_conv(x, w, None, bf)
        def bwd():
            x.grad = w.grad = None
            y.backward(go, retain_graph=True)
        b = t(bwd)
        row += f"{name} fwd {f:7.1f} us ({fl/f/1e6:6.1f} TF/s) bwd {b:7.1f} us ({2*fl/b/1e6:6.1f} TF/s) | "
    wt = w[..., 0]
    lib = t(lambda: torch.matmul(wt, x))
    row += f"rocBLAS f32 fwd {lib:7.1f} us ({fl/lib/1e6:6.1f} TF/s)"
    print(row)
