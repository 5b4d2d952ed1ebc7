"""Masked refinement of AMContrast3D++ (RefinementMethod.DualMasks, fusion 'MIN'; openpoints/AMContrast3D/MaskedRefine.py:55-131)
on csrc/refine.hip against the tensor-operation form it replaces (the mirror of the reference's code, kept behind
AMC3D_NO_FUSED_REFINE=1): forward bit for bit, backward up to the order of the float atomics / index_add."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, D, n, K, gamma, seed):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, n, 3, generator=g).to(DEV)
    f = torch.randn(B, D, n, generator=g).to(DEV)
    a = torch.rand(B * n, 1, generator=g).to(DEV)
    a[torch.rand(B * n, generator=g).to(DEV) < 0.1] = 0.95   # ties among the neighbours' ambiguities
    xyz = p.view(-1, 3).contiguous()
    o = torch.tensor([B * n], dtype=torch.int32, device=DEV)
    idx, _ = ops.knnquery(K, xyz, xyz, o, o)
    return p, f, a, idx[:, 1:].contiguous(), torch.randn(B, D, n, generator=g).to(DEV)


@pytest.mark.parametrize("B,D,n,K,gamma", [(2, 32, 1000, 12, 1.0), (2, 64, 1500, 12, 0.7), (1, 20, 333, 5, 1.0), (3, 256, 94, 12, 0.5),
                                           (2, 128, 376, 12, 1.0)])
def test_fused_refinement_equals_tensor_form(B, D, n, K, gamma, monkeypatch):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.AMContrast3D.MaskedRefine import RefinementMethod
    p, f, a, nidx, go = _case(B, D, n, K, gamma, D + n)
    res = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("AMC3D_NO_FUSED_REFINE", "1")
        fr = f.clone().requires_grad_(True)
        stage_list = {'geometry': {'refine': {-1: nidx}}}
        r = RefinementMethod(stage_list, p, fr, a.unsqueeze(0).view(B, 1, -1), -1, B, K, 'MIN', 1.0, 0.9, gamma)
        out, rate = r.DualMasks()
        out.backward(go)
        res.append((out.detach(), float(rate), fr.grad.clone()))
    (o1, r1, g1), (o0, r0, g0) = res
    assert torch.equal(o1, o0)
    assert abs(r1 - r0) <= 1e-4 and 0.0 < r1 < 100.0
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    assert int((o0 != f).any(1).sum()) > 0  # something was refined


def test_fused_refinement_is_graph_safe():
    from amcontrast3d_amd import ops
    p, f, a, nidx, go = _case(2, 64, 1500, 12, 1.0, 7)
    fs = f.clone().requires_grad_(True)
    out, count = ops.MaskedRefineDual.apply(fs, a, nidx, 0.9, 1.0, 1.0)
    (grad,) = torch.autograd.grad(out, fs, go)
    # a fresh leaf for the captured part: the gradient accumulator of `fs` was created on the default stream by the eager
    # backward above, and autograd would synchronise with that stream inside the capture
    fs = f.clone().requires_grad_(True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            o2, c2 = ops.MaskedRefineDual.apply(fs, a, nidx, 0.9, 1.0, 1.0)
            (g2,) = torch.autograd.grad(o2, fs, go)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        o2, c2 = ops.MaskedRefineDual.apply(fs, a, nidx, 0.9, 1.0, 1.0)
        (g2,) = torch.autograd.grad(o2, fs, go)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(o2, out) and int(c2) == int(count)
    assert float((g2 - grad).abs().max()) <= 1e-5 * float(grad.abs().max())
