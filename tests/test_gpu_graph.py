"""The C-ABI calls are claimed graph-capturable (include/amc3d.h: no allocation, no sync): capture the
data-dependent ones in a hipGraph, replay many times and require bit-identical results every time.
(Regression: with hipMemsetAsync nodes inside the k-NN grid build, replays intermittently used stale
cell counters; the library now zero-fills with ordinary kernels.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_knn_and_ambiguity_under_graph_replay():
    from amcontrast3d_amd import ops, synthetic
    p = torch.from_numpy(synthetic.make_batch(4, 3000, first_id=50)["pos"].reshape(-1, 3)).to(DEV)
    lab = torch.from_numpy(synthetic.make_batch(4, 3000, first_id=50)["y"].reshape(-1)).to(DEV).to(torch.int32)
    o = torch.tensor([p.shape[0]], dtype=torch.int32, device=DEV)

    def run():
        idx, dist = ops.knnquery(24, p, p, o, o)  # 12000^2 pairs: grid path
        nidx = idx[:, 1:]
        pm = ops.posmask_from_labels(lab, nidx)
        a = ops.ambiguity(p, pm, nidx, "Method2", 0.04)
        return idx, dist, a

    want = [t.clone() for t in run()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = run()
    for it in range(25):
        for t in got:
            t.fill_(-1) if t.dtype != torch.float32 else t.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]), it
        assert torch.equal(got[1], want[1]), it
        assert torch.equal(got[2], want[2]), it
