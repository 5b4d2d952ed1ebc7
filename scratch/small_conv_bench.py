"""the skip convs of SA2-4 (1x1, with bias, few positions): this library's kernels vs torch (MIOpen), fwd and bwd"""
import sys, torch
sys.path.insert(0, '.')
from amcontrast3d_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for B, ci, co, P in [(8, 64, 128, 1500), (8, 128, 256, 375), (8, 256, 512, 93), (8, 32, 64, 6000)]:
    x = torch.randn(B, ci, P, device=dev, requires_grad=True)
    conv = torch.nn.Conv1d(ci, co, 1).to(dev)
    go = torch.randn(B, co, P, device=dev)
    row = f"B{B} {ci}->{co} P{P}: "
    for name, f in (("amc", lambda: ops.pointwise_conv(x, conv.weight, conv.bias)), ("torch", lambda: conv(x))):
        tf = t(f)
        y = f()
        def bwd():
            x.grad = None; conv.weight.grad = None; conv.bias.grad = None
            y.backward(go, retain_graph=True)
        tb = t(bwd)
        row += f"{name} fwd {tf:6.1f} bwd {tb:6.1f} us | "
    print(row)
