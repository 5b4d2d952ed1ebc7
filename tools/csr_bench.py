"""GroupedConvBN forward + backward over reverse lists at the shapes of PointNeXt-S' SetAbstraction 1 and 2 (B = 8), a few
iterations: run under `rocprofv3 --kernel-trace --stats` and read csr_collapse_kernel's average (AMC3D_CSR_SHAPE picks the
workgroup shape, csrc/csr.hip).  Also checks that the gradients of the chosen shape equal shape 0's bits when given a file.

    AMC3D_CSR_SHAPE=1 python tools/csr_bench.py [--save ref.pt | --check ref.pt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--check")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for tag, (B, Cin, C, N, M, K) in {"sa1": (8, 32, 32, 24000, 6000, 32), "sa2": (8, 64, 64, 6000, 1500, 32)}.items():
        g = torch.Generator().manual_seed(5)
        # neighbourhoods as a ball query gives them: a centroid's K picks come from a window of nearby ids, with repeats
        centre = torch.randint(0, N, (B, M, 1), generator=g)
        idx = ((centre + torch.randint(-40, 41, (B, M, K), generator=g)) % N).to(torch.int32).to(dev)
        dp = (torch.randn(B, 3, M, K, generator=g) * 0.05).to(dev)
        f = torch.randn(B, Cin, N, generator=g).to(dev)
        w = (torch.randn(C, Cin + 3, 1, 1, generator=g) * 0.2).to(dev)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        go = torch.randn(B, C, M, K, generator=g).to(dev)
        start, edge = ops.group_csr(idx, N)
        edge_dp = ops.group_csr_dp(idx, dp, edge)
        mom = ops.group_moments_csr(idx, dp, N, (start, edge))
        for it in range(a.iters):
            fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
            x1 = ops.GroupedConvBN.apply(fr, dp, idx, mom, wr, gr, br, 1e-5, True, None, (start, edge, edge_dp))
            x1.backward(go)
        torch.cuda.synchronize()
        out[tag] = [t.grad.cpu() for t in (fr, wr, gr, br)]
    if a.save:
        torch.save(out, a.save)
    if a.check:
        ref = torch.load(a.check, weights_only=True)
        worst = 0.0
        for tag in out:
            for x, y in zip(out[tag], ref[tag]):
                # (the fp64 partial sums are grouped by workgroup: another shape may round the last bit of a sum differently)
                d = float((x - y).abs().max()) / max(1e-30, float(y.abs().max()))
                assert d <= 1e-6, (tag, d)
                worst = max(worst, d)
        print("gradients against", a.check, "worst relative difference", worst)
    print("csr_bench done, shape", os.environ.get("AMC3D_CSR_SHAPE", "0"))


if __name__ == "__main__":
    main()
