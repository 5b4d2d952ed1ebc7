"""Fused training-mode BatchNorm (+ReLU, + neighbourhood max) kernels against torch's own layers
(nn.BatchNorm2d / ReLU / torch.max with autograd) on the same GPU tensors, fp64 as the arbiter."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,relu", [((2, 32, 500, 32), True), ((3, 7, 65, 31), False), ((8, 64, 3000), True),
                                        ((2, 5, 1), False), ((2, 16, 33, 4), True),
                                        # one workgroup per channel, one launch (>= 64 channels, <= 16 k elements each); larger ones: two launches
                                        ((8, 256, 375), True), ((8, 128, 1500), False), ((3, 70, 333), True),
                                        ((8, 64, 6000), True), ((2, 64, 93, 32), True)])
def test_bn_act_matches_torch(shape, relu):
    from amcontrast3d_amd.ops import BatchNormAct
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(shape, generator=g) * 3 + 1.5).to(DEV)
    C = shape[1]
    gamma = (torch.rand(C, generator=g) - 0.3).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    go = torch.randn(shape, generator=g).to(DEV)

    def ref(dtype):
        xr = x.to(dtype).requires_grad_(True)
        gr, br = gamma.to(dtype).requires_grad_(True), beta.to(dtype).requires_grad_(True)
        y = torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
        if relu:
            y = torch.relu(y)
        y.backward(go.to(dtype))
        return y.detach(), xr.grad, gr.grad, br.grad

    xg = x.clone().requires_grad_(True)
    gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y, mean, var_u = BatchNormAct.apply(xg, gg, bg, 1e-5, relu)
    y.backward(go)
    y64, dx64, dg64, db64 = ref(torch.float64)
    y32, dx32, dg32, db32 = ref(torch.float32)
    for got, r64, r32 in ((y, y64, y32), (xg.grad, dx64, dx32), (gg.grad, dg64, dg32), (bg.grad, db64, db32)):
        err = float((got.double() - r64).abs().max())
        err_torch = float((r32.double() - r64).abs().max())
        scale = max(1.0, float(r64.abs().max()))
        assert err <= max(2 * err_torch, 1e-5 * scale), (err, err_torch)
    dims = [0] + list(range(2, x.dim()))
    assert torch.allclose(mean, x.mean(dims), atol=1e-5)
    if x.numel() // C > 1:
        assert torch.allclose(var_u, x.var(dims, unbiased=True), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 32, 24000), (3, 7, 65), (8, 128, 1500), (2, 64, 6001), (2, 16, 33, 4)])
def test_bn_residual_act_matches_torch(shape):
    """relu(bn(x) + res): the tail of an InvResMLP block (pointnext_AA.py:296-307) in the BatchNorm kernels"""
    from amcontrast3d_amd.ops import BatchNormResidualAct
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(shape, generator=g) * 3 + 1.5).to(DEV)
    res = torch.randn(shape, generator=g).to(DEV)
    C = shape[1]
    gamma = (torch.rand(C, generator=g) - 0.3).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    go = torch.randn(shape, generator=g).to(DEV)

    def ref(dtype):
        leaves = [t.to(dtype).requires_grad_(True) for t in (x, res, gamma, beta)]
        y = torch.relu(torch.nn.functional.batch_norm(leaves[0], None, None, leaves[2], leaves[3], True, 0.1, 1e-5) + leaves[1])
        y.backward(go.to(dtype))
        return [y.detach()] + [t.grad for t in leaves]

    leaves = [t.clone().requires_grad_(True) for t in (x, res, gamma, beta)]
    y, mean, var_u = BatchNormResidualAct.apply(leaves[0], leaves[1], leaves[2], leaves[3], 1e-5)
    y.backward(go)
    r64, r32 = ref(torch.float64), ref(torch.float32)
    for got, a64, a32 in zip([y] + [t.grad for t in leaves], r64, r32):
        err = float((got.double() - a64).abs().max())
        err_torch = float((a32.double() - a64).abs().max())
        scale = max(1.0, float(a64.abs().max()))
        assert err <= max(2 * err_torch, 1e-5 * scale), (err, err_torch)
    dims = [0] + list(range(2, x.dim()))
    assert torch.allclose(mean, x.mean(dims), atol=1e-5)


@pytest.mark.parametrize("shape", [(8, 32, 6000), (8, 1, 1500), (2, 2, 93), (3, 16, 375), (8, 4, 24000)])
def test_bn_sigmoid_matches_torch(shape):
    """sigmoid(bn(x)): BatchNorm1d -> Sigmoid of the APM towers (APM/concatenation.py:20-60)"""
    from amcontrast3d_amd.ops import BatchNormSigmoid
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV)
    C = shape[1]
    gamma = (torch.rand(C, generator=g) + 0.3).to(DEV)
    beta = torch.randn(C, generator=g).to(DEV)
    go = torch.randn(shape, generator=g).to(DEV)

    def ref(dtype):
        leaves = [t.to(dtype).requires_grad_(True) for t in (x, gamma, beta)]
        y = torch.sigmoid(torch.nn.functional.batch_norm(leaves[0], None, None, leaves[1], leaves[2], True, 0.1, 1e-5))
        y.backward(go.to(dtype))
        return [y.detach()] + [t.grad for t in leaves]

    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta)]
    y, mean, var_u = BatchNormSigmoid.apply(leaves[0], leaves[1], leaves[2], 1e-5)
    y.backward(go)
    r64, r32 = ref(torch.float64), ref(torch.float32)
    for got, a64, a32 in zip([y] + [t.grad for t in leaves], r64, r32):
        err = float((got.double() - a64).abs().max())
        err_torch = float((a32.double() - a64).abs().max())
        scale = max(1.0, float(a64.abs().max()))
        assert err <= max(2 * err_torch, 1e-5 * scale), (err, err_torch)


def test_invresmlp_fused_residual_equals_unfused(monkeypatch):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import timing
    from openpoints.models.backbone.pointnext_AA import InvResMLP
    from openpoints.utils import EasyConfig
    torch.manual_seed(0)
    ga = EasyConfig(); ga.update({'NAME': 'ballquery', 'radius': 0.2, 'nsample': 32})
    blk = InvResMLP(32, norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, group_args=ga, expansion=4).to(DEV).train()
    p = torch.rand(2, 1500, 3, device=DEV)
    f = torch.randn(2, 32, 1500, device=DEV)
    go = torch.randn(2, 32, 1500, device=DEV)
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def run():
        blk.load_state_dict(state)
        for q in blk.parameters():
            q.grad = None
        fi = f.clone().requires_grad_(True)
        with timing.count_calls() as calls:
            out = blk([p, fi])[1]
            out.backward(go)
        return out.detach(), fi.grad, [q.grad.clone() for q in blk.parameters()], dict(calls)

    o1, d1, g1, c1 = run()
    monkeypatch.setenv("AMC3D_NO_BN_RESIDUAL", "1")
    o0, d0, g0, c0 = run()
    assert c1.get("bn_residual_forward") == 1 and c1.get("bn_residual_backward") == 1 and "bn_residual_forward" not in c0, (c1, c0)
    assert float((o1 - o0).abs().max()) <= 1e-5 * float(o0.abs().max())
    assert float((d1 - d0).norm()) <= 1e-4 * float(d0.norm())
    for a, b in zip(g1, g0):
        assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-7


@pytest.mark.parametrize("shape,relu", [((2, 32, 500, 32), False), ((2, 16, 100, 32), True), ((3, 9, 17, 5), False)])
def test_bn_max_matches_torch(shape, relu):
    from amcontrast3d_amd.ops import BatchNormMax
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(shape, generator=g) * 2 - 0.5).to(DEV)
    x[:, :, :, -1] = x[:, :, :, 0]  # duplicated neighbours (ball-query padding): ties in the max
    C = shape[1]
    gamma = (torch.rand(C, generator=g) - 0.4).to(DEV)  # some negative gammas
    beta = torch.randn(C, generator=g).to(DEV)
    go = torch.randn(shape[:3], generator=g).to(DEV)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if relu:
        y = torch.relu(y)
    y = y.max(-1)[0]
    y.backward(go.double())
    xg = x.clone().requires_grad_(True)
    gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    out, _, _ = BatchNormMax.apply(xg, gg, bg, 1e-5, relu)
    out.backward(go)
    assert float((out.double() - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))
    # the dense part of dx and the parameter gradients do not depend on which of two tied neighbours is taken
    assert float((gg.grad.double() - gr.grad).abs().max()) <= 1e-4 * max(1.0, float(gr.grad.abs().max()))
    assert float((bg.grad.double() - br.grad).abs().max()) <= 1e-4 * max(1.0, float(br.grad.abs().max()))
    a, b = xg.grad.double(), xr.grad
    pair = lambda t: torch.cat([t[..., 1:-1], (t[..., :1] + t[..., -1:])], -1)  # tied columns summed
    assert float((pair(a) - pair(b)).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("momentum", [0.1, None, 0.37])
def test_run_convblocks_equals_module_stack_and_updates_running_stats(momentum):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models.layers import create_convblock2d, run_convblocks
    torch.manual_seed(0)
    norm = {'norm': 'bn', 'momentum': momentum}
    blocks = nn.Sequential(create_convblock2d(7, 16, norm_args=norm, act_args={'act': 'relu'}),
                           create_convblock2d(16, 24, norm_args=norm, act_args=None)).to(DEV).train()
    import copy
    ref = copy.deepcopy(blocks)
    x = torch.randn(2, 7, 300, 32, device=DEV)
    for _ in range(2):  # two steps: the cumulative average (momentum None) depends on the step count
        got = run_convblocks(blocks, x, pool_max=True)
        want = ref(x).max(-1)[0]
    assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    for (k, a), (_, b) in zip(blocks.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), k
    blocks.eval(); ref.eval()
    assert torch.allclose(run_convblocks(blocks, x, pool_max=True), ref(x).max(-1)[0], atol=1e-5)
