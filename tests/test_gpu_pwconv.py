"""Pointwise (1x1) convolution on fp32 MFMA (csrc/pwconv.hip) against torch's conv1d/conv2d, forward and
backward, with an fp64 evaluation as the arbiter: the kernel may be no further from fp64 than a few times
what torch's own fp32 kernel is."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [  # B, Cin, Cout, spatial, bias
    (2, 4, 32, (1000,), False),      # stem
    (2, 32, 13, (777,), True),       # head: bias, Cout not a multiple of 32, P % 4 != 0
    (8, 96, 32, (3000,), False),     # FP
    (2, 131, 256, (375, 32), False),  # SA3 (Conv2d)
    (1, 259, 512, (93, 32), False),   # SA4
    (3, 768, 256, (94,), False),     # deepest FP, ragged P
    (1, 1, 1, (5,), True),
    (2, 64, 64, (129,), False),
    # deep layers with P % 4 == 0: the streaming weight-gradient kernel (csrc/gemm.hip gw_wgrad_kernel), 64- and 128-wide tiles
    (2, 64, 128, (300, 32), False),   # SA2 of PointNeXt-S
    (1, 128, 256, (100, 32), False),  # SA3
    (3, 256, 512, (23, 32), False),   # SA4: short position ranges
    (2, 96, 200, (1000,), False),     # ragged channel counts
    (2, 64, 64, (4000,), False),
    # short deep layers: K split over workgroups, bias added by the reduction
    (8, 256, 512, (94,), True),       # SA4 skip conv
    (8, 128, 256, (375,), True),
    (2, 768, 256, (96,), False),
    (2, 256, 96, (200,), False),      # cout <= 128 but short: backward-data on the tiled GEMM as well
    # the wide, short layers of PointNeXt-XL (InvResMLP 512 -> 2048 over 2 x 248 positions): a handful of partials per
    # weight gradient, summed by the few-partials reduction
    (2, 512, 2048, (248,), False),
    (2, 1024, 256, (64,), False),
]


@pytest.mark.parametrize("B,Cin,Cout,spatial,bias", SHAPES)
def test_pointwise_conv_matches_torch(B, Cin, Cout, spatial, bias):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(B, Cin, *spatial, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, *([1] * len(spatial)), generator=g) * 0.1).to(DEV)
    bvec = torch.randn(Cout, generator=g).to(DEV) if bias else None
    go = torch.randn(B, Cout, *spatial, generator=g).to(DEV)
    conv = F.conv1d if len(spatial) == 1 else F.conv2d

    def ref(dtype):
        xr, wr = x.to(dtype).requires_grad_(True), w.to(dtype).requires_grad_(True)
        br = bvec.to(dtype).requires_grad_(True) if bias else None
        y = conv(xr, wr, br)
        y.backward(go.to(dtype))
        return y.detach(), xr.grad, wr.grad, (br.grad if bias else None)

    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bg = bvec.clone().requires_grad_(True) if bias else None
    y = ops.pointwise_conv(xg, wg, bg)
    assert y.shape == go.shape
    y.backward(go)
    r64, r32 = ref(torch.float64), ref(torch.float32)
    got = (y, xg.grad, wg.grad, bg.grad if bias else None)
    for a, b64, b32, what in zip(got, r64, r32, ("y", "dx", "dw", "db")):
        if a is None:
            continue
        assert a.shape == b64.shape, what
        err = float((a.double() - b64).abs().max())
        err_torch = float((b32.double() - b64).abs().max())
        scale = max(1.0, float(b64.abs().max()))
        assert err <= max(4 * err_torch, 2e-6 * scale), (what, err, err_torch, scale)


def test_pointwise_conv_weight_gradient_is_deterministic():
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 96, 5000, generator=g).to(DEV)
    w = (torch.randn(32, 96, 1, generator=g) * 0.1).to(DEV)
    go = torch.randn(4, 32, 5000, generator=g).to(DEV)
    grads = []
    for _ in range(3):
        wg = w.clone().requires_grad_(True)
        ops.pointwise_conv(x, wg).backward(go)
        grads.append(wg.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def test_model_blocks_route_1x1_convs_to_the_kernel():
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import timing
    from openpoints.models.layers import create_convblock1d, run_convblocks
    blk = torch.nn.Sequential(create_convblock1d(8, 16, norm_args={'norm': 'bn'}, act_args={'act': 'relu'}),
                              create_convblock1d(16, 5, norm_args=None, act_args=None)).to(DEV)
    x = torch.randn(2, 8, 20000, device=DEV)  # wide enough for the shape heuristic (blocks._pw_pays)
    timing.enable(True)
    try:
        y = run_convblocks(blk, x)
        torch.cuda.synchronize()
        names = set(timing.collect().keys())
    finally:
        timing.enable(False)
    assert "pointwise_conv_forward" in names, names
    blk.train()
    ref = blk(x)
    assert torch.allclose(y, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("shape", [(8, 256, 256, (375,)), (4, 768, 256, (94,)), (2, 128, 192, (50, 32))])
def test_library_gemm_conv_matches_torch_conv(shape):
    """deep, short layers: the three-GEMM decomposition (ops.LibraryGemmConv) against nn.functional.conv, in both
    weight-gradient forms (the form is a pure function of the shape; AMC3D_WGRAD_FORM overrides it)"""
    import os
    from amcontrast3d_amd import ops
    B, ci, co, sp = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, ci, *sp, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(co, ci, *([1] * len(sp)), generator=g) * 0.05).to(dev).requires_grad_(True)
    go = torch.randn(B, co, *sp, generator=g).to(dev)
    conv = torch.nn.functional.conv1d if len(sp) == 1 else torch.nn.functional.conv2d
    yr = conv(x, w)
    yr.backward(go)
    want = (yr.detach(), x.grad.clone(), w.grad.clone())
    for form in ("", "bmm", "flat"):
        os.environ["AMC3D_WGRAD_FORM"] = form
        x.grad = w.grad = None
        y = ops.library_gemm_conv(x, w)
        y.backward(go)
        for got, ref in zip((y.detach(), x.grad, w.grad), want):
            assert got.shape == ref.shape
            assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    os.environ.pop("AMC3D_WGRAD_FORM", None)


@pytest.mark.parametrize("B,ci,co,P", [(2, 64, 256, 3000), (1, 35, 32, 5003), (3, 256, 64, 777), (2, 512, 512, 186),
                                      (1, 1024, 256, 93), (4, 4, 32, 4096)])
def test_pointwise_conv_bf16_compute(B, ci, co, P):
    """csrc/gemm_bf16.hip: operands rounded to bf16 (round-to-nearest-even), exact products, fp32 accumulation.  Checked
    against torch in fp64 ON THE SAME bf16-ROUNDED OPERANDS (then only the fp32 summation order differs: 1e-5), and
    against the unrounded fp32 product at the bf16 bound (2^-8 per operand, sqrt(K) growth)."""
    from amcontrast3d_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, ci, P, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(co, ci, 1, generator=g) * 0.1).to(dev).requires_grad_(True)
    bias = torch.randn(co, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(B, co, P, generator=g).to(dev)
    y = ops.pointwise_conv(x, w, bias, True)
    y.backward(go)
    r = lambda t: t.detach().to(torch.bfloat16).double()
    yr = torch.einsum("oc,bcp->bop", r(w)[..., 0], r(x)) + bias.detach().double()[None, :, None]
    dxr = torch.einsum("oc,bop->bcp", r(w)[..., 0], r(go))
    dwr = torch.einsum("bop,bcp->oc", r(go), r(x))
    for name, got, ref in (("y", y, yr), ("dx", x.grad, dxr), ("dw", w.grad[..., 0], dwr), ("db", bias.grad, go.double().sum((0, 2)))):
        err = float((got.double() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (name, err)
    y32 = torch.einsum("oc,bcp->bop", w.detach().double()[..., 0], x.detach().double()) + bias.detach().double()[None, :, None]
    assert float((y.double() - y32).abs().max()) <= 2 ** -7 * float(y32.abs().max()) + 1e-3


@pytest.mark.parametrize("rows,c1,c2", [(64, 3, 32), (512, 128, 256), (5, 1, 7), (32, 3, 0 + 61)])
def test_split_and_join_columns(rows, c1, c2):
    """amc3d_split_columns / amc3d_join_columns: the [W_dp | W_f] and [W_skip | W_up] weight blocks, one launch each way"""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(rows + c1)
    w = torch.randn(rows, c1 + c2, generator=g).to(DEV)
    a, b = ops._split_columns(w, c1)
    assert torch.equal(a, w[:, :c1]) and torch.equal(b, w[:, c1:]) and a.is_contiguous() and b.is_contiguous()
    assert torch.equal(ops._join_columns(a, b), w)
    wl = w.clone().reshape(rows, c1 + c2, 1).requires_grad_(True)
    x, y = ops.split_weight(wl, c1)
    (x.sum() * 2 + (y * y).sum()).backward()
    want = torch.cat((torch.full_like(a, 2.0), 2 * b), 1).view(rows, c1 + c2, 1)
    assert torch.equal(wl.grad, want)


@pytest.mark.parametrize("shape", [(8, 32, 24000), (2, 13, 777), (3, 1, 5), (1, 257, 96)])
def test_bias_grad(shape):
    import ctypes
    from amcontrast3d_amd import _lib
    g = torch.Generator().manual_seed(shape[1])
    dy = torch.randn(shape, generator=g).to(DEV)
    db = torch.empty(shape[1], device=DEV)
    _lib.check(_lib.load().amc3d_bias_grad(shape[0], shape[1], shape[2], ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(db.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "bias_grad")
    want = dy.double().sum((0, 2))
    assert float((db.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    db2 = torch.empty_like(db)
    _lib.check(_lib.load().amc3d_bias_grad(shape[0], shape[1], shape[2], ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(db2.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "bias_grad")
    assert torch.equal(db, db2)  # fixed summation order
