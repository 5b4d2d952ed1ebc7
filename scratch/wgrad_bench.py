"""weight gradient of the small deep 1x1 layers: csrc/gemm.hip (split-K) vs library GEMM forms, graph-timed"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from amcontrast3d_amd import _lib, ops
DEV = 'cuda:0'
lib = _lib.load()
shapes = [(8, 259, 256, 2976), (8, 256, 512, 2976), (8, 768, 256, 375), (8, 256, 256, 375), (8, 384, 128, 1500),
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
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
for B, Ci, Co, P in shapes:
    x = torch.randn(B, Ci, P, device=DEV); w = torch.randn(Co, Ci, device=DEV); go = torch.randn(B, Co, P, device=DEV)
    dw = torch.empty(Co, Ci, device=DEV)
    wb = int(lib.amc3d_pointwise_conv_workspace_bytes(B, Ci, Co, P)); ws = torch.empty(wb, dtype=torch.uint8, device=DEV)
    def mine():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.amc3d_pointwise_conv_backward(B, Ci, Co, P, P_(x), P_(w), P_(go), None, P_(dw), P_(ws), wb, st), "bw")
    def mine_d():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        dx = torch.empty_like(x)
        _lib.check(lib.amc3d_pointwise_conv_backward(B, Ci, Co, P, P_(x), P_(w), P_(go), P_(dx), None, None, 0, st), "bw")
    t = [tm(mine), tm(mine_d), tm(lambda: torch.bmm(go, x.transpose(1, 2)).sum(0)), tm(lambda: torch.matmul(w.t(), go))]
    ref = torch.einsum('bop,bip->oi', go.double(), x.double()); mine()
    print(f"{Ci:4d}->{Co:4d} P={P:6d}: mine wgrad {t[0]:6.1f} (err {float((dw.double()-ref).abs().max()/ref.abs().max()):.1e}) mine dgrad {t[1]:6.1f} | bmm+sum {t[2]:6.1f} lib dgrad {t[3]:6.1f}", flush=True)
