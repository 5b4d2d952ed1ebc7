"""Recomputing SetAbstraction tail (csrc/sa_tail.hip) against the layer-by-layer torch evaluation it replaces:
BN1 -> ReLU -> Conv2d 1x1 -> BN2 [-> ReLU] -> max over the 32 neighbours, forward and backward, fp64 as the arbiter."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference(y1, g1, b1, w2, g2, b2, relu2, gout, dtype):
    y = y1.to(dtype).requires_grad_(True)
    ps = [t.to(dtype).requires_grad_(True) for t in (g1, b1, w2, g2, b2)]
    x1 = F.relu(F.batch_norm(y, None, None, ps[0], ps[1], True, 0.1, 1e-5))
    z = F.batch_norm(F.conv2d(x1, ps[2]), None, None, ps[3], ps[4], True, 0.1, 1e-5)
    if relu2:
        z = F.relu(z)
    out = z.max(-1)[0]
    out.backward(gout.to(dtype))
    return [out.detach(), y.grad] + [p.grad for p in ps]


@pytest.mark.parametrize("B,C1,C2,M,relu2", [(2, 32, 64, 257, True), (1, 64, 128, 100, False), (2, 16, 32, 33, True),
                                              (1, 32, 96, 7, False), (3, 64, 50, 64, True)])
def test_sa_tail_matches_layerwise_torch(B, C1, C2, M, relu2):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(C1 * 7 + C2)
    y1 = torch.randn(B, C1, M, 32, generator=g).to(DEV) * 2 + 0.3
    g1, b1 = (torch.rand(C1, generator=g) + 0.5).to(DEV), (torch.randn(C1, generator=g) * 0.2).to(DEV)
    w2 = (torch.randn(C2, C1, 1, 1, generator=g) * 0.2).to(DEV)
    g2, b2 = (torch.rand(C2, generator=g) + 0.5).to(DEV), (torch.randn(C2, generator=g) * 0.2).to(DEV)
    gout = torch.randn(B, C2, M, generator=g).to(DEV)
    assert ops.sa_tail_supported(C1, C2, 32)
    leaves = [t.clone().requires_grad_(True) for t in (y1, g1, b1, w2, g2, b2)]
    out = ops.SATail.apply(leaves[0], leaves[1], leaves[2], 1e-5, leaves[3], leaves[4], leaves[5], 1e-5, relu2)
    out.backward(gout)
    got = [out.detach()] + [t.grad for t in leaves]
    r64 = reference(y1, g1, b1, w2, g2, b2, relu2, gout, torch.float64)
    r32 = reference(y1, g1, b1, w2, g2, b2, relu2, gout, torch.float32)
    for a, b64, b32, what in zip(got, r64, r32, ("pooled", "dy1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2")):
        assert a.shape == b64.shape, what
        err = float((a.double() - b64).abs().max())
        err_torch = float((b32.double() - b64).abs().max())
        scale = max(1.0, float(b64.abs().max()))
        # arg-max flips between near-equal neighbours move single gradient entries: compare norm-wise for gradients
        if what == "pooled":
            assert err <= max(4 * err_torch, 2e-6 * scale), (what, err, err_torch)
        else:
            rel = float((a.double() - b64).norm() / (b64.norm() + 1e-30))
            rel_torch = float((b32.double() - b64).norm() / (b64.norm() + 1e-30))
            assert rel <= max(4 * rel_torch, 1e-5), (what, rel, rel_torch)


def test_sa_blocks_route_through_the_tail_and_match_the_module_stack():
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    import copy
    from amcontrast3d_amd import timing
    from openpoints.models.layers import create_convblock2d, run_convblocks
    torch.manual_seed(1)
    blocks = nn.Sequential(create_convblock2d(35, 32, norm_args={'norm': 'bn'}, act_args={'act': 'relu'}),
                           create_convblock2d(32, 64, norm_args={'norm': 'bn'}, act_args=None)).to(DEV).train()
    ref = copy.deepcopy(blocks)
    x = torch.randn(2, 35, 150, 32, device=DEV)
    pre = blocks[0][0](x)  # what the fused gather+conv kernel hands over
    timing.enable(True)
    try:
        got = run_convblocks(blocks, None, pool_max=True, pre=pre)
        torch.cuda.synchronize()
        names = set(timing.collect().keys())
    finally:
        timing.enable(False)
    assert "sa_tail_forward" in names, names
    want = ref(x).max(-1)[0]
    assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    for (k, a), (_, b) in zip(blocks.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), k  # running stats of both BatchNorms


@pytest.mark.parametrize("B,C1,C2,M", [(2, 32, 64, 257), (1, 64, 128, 100), (2, 16, 32, 33)])
def test_activated_tail_hands_its_gradient_over_position_major(B, C1, C2, M, monkeypatch):
    """ops.SATailActivated writes d/dx1 as (B,M,32,C1) rows behind a (B,C1,M,32) view -- the layout the first layer's
    gathering backward reads -- with the same values as the channel-major store"""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(C1 + C2 + M)
    x1 = torch.relu(torch.randn(B, C1, M, 32, generator=g)).to(DEV)
    w2 = (torch.randn(C2, C1, 1, 1, generator=g) * 0.2).to(DEV)
    g2, b2 = (torch.rand(C2, generator=g) + 0.5).to(DEV), (torch.randn(C2, generator=g) * 0.2).to(DEV)
    gout = torch.randn(B, C2, M, generator=g).to(DEV)
    res = []
    for cm in (True, False):
        if cm:
            monkeypatch.setenv("AMC3D_SAT_DX1_CM", "1")
        else:
            monkeypatch.delenv("AMC3D_SAT_DX1_CM")
        leaves = [t.clone().requires_grad_(True) for t in (x1, w2, g2, b2)]
        seen = []
        leaves[0].register_hook(lambda gr: seen.append(gr))  # the gradient tensor as the producer of x1 receives it
        out = ops.SATailActivated.apply(leaves[0], leaves[1], leaves[2], leaves[3], 1e-5, True, None)
        out.backward(gout)
        res.append((seen[0], [t.grad for t in leaves[1:]], out.detach()))
    (d_cm, r_cm, o_cm), (d_pm, r_pm, o_pm) = res
    assert d_cm.is_contiguous() and d_pm.permute(0, 2, 3, 1).is_contiguous() and d_pm.shape == d_cm.shape
    assert torch.equal(d_pm, d_cm) and torch.equal(o_cm, o_pm)
    assert all(torch.equal(a, b) for a, b in zip(r_cm, r_pm))
