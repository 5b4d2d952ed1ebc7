"""Fused gather + concat + 1x1 conv (fp32 MFMA) against the reference's chain
grouping_operation -> torch.cat([dp, fj]) -> F.conv2d, forward and backward, fp64 as the arbiter."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,C,Cout,N,M,K", [(2, 32, 32, 1000, 250, 32), (1, 64, 64, 700, 300, 32), (2, 32, 64, 513, 77, 32),
                                            (1, 16, 32, 300, 129, 16), (2, 64, 128, 400, 100, 8), (1, 5, 96, 50, 7, 3)])
def test_grouped_conv_matches_reference_chain(B, C, Cout, N, M, K):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C)
    f = torch.randn(B, C, N, generator=g).to(DEV)
    dp = torch.randn(B, 3, M, K, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, M, K), generator=g, dtype=torch.int32).to(DEV)
    idx[:, :, -1] = idx[:, :, 0]  # repeated neighbours, as ball-query padding produces
    w = (torch.randn(Cout, C + 3, 1, 1, generator=g) * 0.2).to(DEV)
    go = torch.randn(B, Cout, M, K, generator=g).to(DEV)
    assert ops.grouped_conv_supported(C, Cout)

    def ref(dtype):
        fr, wr = f.to(dtype).requires_grad_(True), w.to(dtype).requires_grad_(True)
        flat = idx.reshape(B, 1, -1).expand(-1, C, -1).long()
        fj = fr.gather(2, flat).reshape(B, C, M, K)
        y = F.conv2d(torch.cat([dp.to(dtype), fj], 1), wr)
        y.backward(go.to(dtype))
        return y.detach(), fr.grad, wr.grad

    fg, wg = f.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.grouped_conv(fg, dp, idx, wg)
    y.backward(go)
    y64, df64, dw64 = ref(torch.float64)
    y32, df32, dw32 = ref(torch.float32)
    for got, r64, r32, what in ((y, y64, y32, "y"), (fg.grad, df64, df32, "df"), (wg.grad, dw64, dw32, "dw")):
        err = float((got.double() - r64).abs().max())
        err_torch = float((r32.double() - r64).abs().max())
        scale = max(1.0, float(r64.abs().max()))
        assert err <= max(4 * err_torch, 2e-6 * scale), (what, err, err_torch, scale)


def test_grouped_conv_unsupported_shapes_fall_back_in_the_model():
    from amcontrast3d_amd import ops
    assert not ops.grouped_conv_supported(128, 128) and not ops.grouped_conv_supported(32, 48)
    assert ops.grouped_conv_supported(32, 32) and ops.grouped_conv_supported(64, 64)
