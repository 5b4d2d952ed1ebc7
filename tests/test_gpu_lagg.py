"""Convolve-before-gather neighbourhood aggregation (csrc/lagg.hip, ops.LocalAggregationFused) against the layer as the
reference writes it -- grouping_operation + cat([dp, fj]) + Conv2d 1x1 + BatchNorm2d (batch statistics) [+ ReLU] + max
over the neighbours (pointnext_AA.py:57-63, 139-170) -- evaluated by torch in fp64 on the same inputs: forward values,
arg-max picks, batch statistics / running buffers, and every gradient (routing = the kernel's own picks)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, Cin, C, N, M, K, seed, relu, radius=0.35):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    p = torch.rand(B, N, 3, generator=g).to(dev)
    q = p[:, :M].contiguous()
    idx = ops.ball_query(radius, K, p, q)
    dp = (ops.grouping_operation(p.transpose(1, 2).contiguous(), idx) - q.transpose(1, 2).unsqueeze(-1)) / radius
    f = torch.randn(B, Cin, N, generator=g).to(dev)
    w = (torch.randn(C, Cin + 3, 1, 1, generator=g) * 0.3).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    gamma[::5] *= -1  # negative scale factors too
    beta = (torch.randn(C, generator=g) * 0.2).to(dev)
    go = torch.randn(B, C, M, generator=g).to(dev)
    return p, idx, dp.contiguous(), f, w, gamma, beta, go


def _reference(idx, dp, f, w, gamma, beta, go, relu, arg, eps=1e-5):
    """fp64 torch evaluation; the max-pool gathers at `arg` (the kernel's picks) so that the gradients are comparable"""
    B, Cin, N = f.shape
    f = f.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    gamma = gamma.double().requires_grad_(True)
    beta = beta.double().requires_grad_(True)
    fj = f.gather(2, idx.reshape(B, 1, -1).expand(-1, Cin, -1).long()).reshape(B, Cin, idx.shape[1], idx.shape[2])
    x = torch.cat([dp.double(), fj], 1)
    y = torch.nn.functional.conv2d(x, w)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    z = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps) * gamma[None, :, None, None] + beta[None, :, None, None]
    if relu:
        z = torch.relu(z)
    own = z.max(-1)
    pooled = z.gather(-1, arg.long().unsqueeze(-1)).squeeze(-1)
    pooled.backward(go.double())
    cnt = y.numel() / y.shape[1]
    return {"pooled": pooled.detach(), "own_max": own.values.detach(), "own_arg": own.indices, "mean": mean.detach(),
            "var_u": (var * cnt / (cnt - 1)).detach(), "df": f.grad, "dw": w.grad, "dgamma": gamma.grad, "dbeta": beta.grad}


@pytest.mark.parametrize("B,Cin,C,N,M,K,relu", [
    (2, 8, 8, 700, 700, 32, True),        # test-width LocalAggregation (M = N)
    (2, 16, 32, 900, 225, 32, True),      # strided SetAbstraction
    (3, 32, 64, 1500, 375, 32, False),    # no ReLU behind the BatchNorm
    (2, 64, 64, 1200, 1200, 32, True),
    (1, 128, 128, 800, 800, 32, True),
    (2, 64, 256, 640, 160, 32, True),     # two channel chunks of 128
    (1, 24, 512, 300, 75, 16, True),      # 16 neighbours, four chunks
])
def test_local_aggregation_matches_the_unfused_layer(B, Cin, C, N, M, K, relu):
    from amcontrast3d_amd import ops
    p, idx, dp, f, w, gamma, beta, go = _case(B, Cin, C, N, M, K, 11 + C, relu)
    bn = torch.nn.BatchNorm2d(C).to(f.device)
    mom = ops.group_moments(idx, dp, N)
    fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
    log = {}
    ops.pool_log(log)
    try:
        pooled = ops.LocalAggregationFused.apply(fr, dp, idx, mom, wr, gr, br, 1e-5, relu, bn)
    finally:
        ops.pool_log(None)
    pooled.backward(go)
    arg = log[0]
    ref = _reference(idx, dp, f, w, gamma, beta, go, relu, arg)
    tol = lambda r: 2e-5 * max(1.0, float(r.abs().max()))
    assert float((pooled.double() - ref["pooled"]).abs().max()) <= tol(ref["pooled"])
    # the kernel's pick attains the true maximum (first index among equals, up to fp32 rounding of the values)
    assert float((ref["pooled"] - ref["own_max"]).abs().max()) <= tol(ref["own_max"])
    assert float((arg.long() != ref["own_arg"]).double().mean()) <= 2e-3
    assert float((bn.running_mean.double() - 0.1 * ref["mean"]).abs().max()) <= 1e-5
    assert float((bn.running_var.double() - (0.9 + 0.1 * ref["var_u"])).abs().max()) <= 1e-5 * max(1.0, float(ref["var_u"].max()))
    assert int(bn.num_batches_tracked) == 1
    for name, got, want in (("df", fr.grad, ref["df"]), ("dw", wr.grad, ref["dw"]), ("dgamma", gr.grad, ref["dgamma"]),
                            ("dbeta", br.grad, ref["dbeta"])):
        err = float((got.double() - want).abs().max())
        assert err <= 5e-5 * max(1.0, float(want.abs().max())), (name, err, float(want.abs().max()))


def test_local_aggregation_eval_mode_and_moments():
    from amcontrast3d_amd import ops
    B, Cin, C, N, M, K = 2, 16, 32, 900, 225, 32
    p, idx, dp, f, w, gamma, beta, go = _case(B, Cin, C, N, M, K, 5, True)
    # moments against torch: in-degree, dp sums, global sums
    mom = ops.group_moments(idx, dp, N)
    raw = mom.cpu().numpy()
    glob = raw[:72].view(np.int64).astype(np.float64) / 2.0 ** 30
    cnt = raw[128:128 + 4 * B * N].view(np.int32).reshape(B, N)
    off = 128 + 4 * B * N + 4 * ((B * N) & 1)
    dsum = raw[off:off + 24 * B * N].view(np.int64).reshape(B, N, 3).astype(np.float64) / 2.0 ** 36
    flat = idx.reshape(B, -1).long().cpu()
    want_cnt = torch.stack([torch.bincount(flat[b], minlength=N) for b in range(B)]).numpy()
    np.testing.assert_array_equal(cnt, want_cnt)
    d = dp.reshape(B, 3, -1).double().cpu()
    want_d = torch.zeros(B, N, 3, dtype=torch.float64)
    for b in range(B):
        want_d[b].index_add_(0, flat[b], d[b].t())
    np.testing.assert_allclose(dsum, want_d.numpy(), atol=1e-7)
    np.testing.assert_allclose(glob[:3], d.sum((0, 2)).numpy(), atol=1e-5)
    np.testing.assert_allclose(glob[3], float((d[:, 0] * d[:, 0]).sum()), rtol=1e-8, atol=1e-5)
    # eval mode: running statistics, no gradient
    bn = torch.nn.BatchNorm2d(C).to(f.device)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.copy_(gamma); bn.bias.copy_(beta)
    bn.eval()
    got = ops.local_aggregation_eval(f, dp, idx, w, bn, True)
    fj = ops.grouping_operation(f, idx)
    want = torch.relu(bn(torch.nn.functional.conv2d(torch.cat([dp, fj], 1), w))).max(-1).values
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,Cin,C,N,M", [(2, 8, 8, 900, 225), (2, 32, 32, 2000, 500), (3, 64, 64, 1200, 300),
                                         (2, 128, 128, 640, 160), (1, 256, 256, 372, 93)])
def test_grouped_conv_bn_first_block(B, Cin, C, N, M):
    """ops.GroupedConvBN (first block of PointNeXt-S' two-layer SetAbstraction MLP: conv before the gather, activation
    materialised) against grouping + cat + Conv2d + BatchNorm2d (batch statistics) + ReLU in fp64: x1, running buffers and
    every gradient for a dense upstream gradient."""
    from amcontrast3d_amd import ops
    K = 32
    p, idx, dp, f, w, gamma, beta, _ = _case(B, Cin, C, N, M, K, 31 + C, True)
    go = torch.randn(B, C, M, K, generator=torch.Generator().manual_seed(3)).to(f.device)
    bn = torch.nn.BatchNorm2d(C).to(f.device)
    mom = ops.group_moments(idx, dp, N)
    fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
    x1 = ops.GroupedConvBN.apply(fr, dp, idx, mom, wr, gr, br, 1e-5, True, bn)
    x1.backward(go)
    f64 = f.double().requires_grad_(True)
    w64, g64, b64 = (t.double().requires_grad_(True) for t in (w, gamma, beta))
    fj = f64.gather(2, idx.reshape(B, 1, -1).expand(-1, Cin, -1).long()).reshape(B, Cin, M, K)
    y = torch.nn.functional.conv2d(torch.cat([dp.double(), fj], 1), w64)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    ref = torch.relu((y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * g64[None, :, None, None]
                     + b64[None, :, None, None])
    ref.backward(go.double())
    assert float((x1.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    cnt = y.numel() / C
    assert float((bn.running_mean.double() - 0.1 * mean).abs().max()) <= 1e-5
    assert float((bn.running_var.double() - (0.9 + 0.1 * var * cnt / (cnt - 1))).abs().max()) <= 1e-5 * max(1.0, float(var.max()))
    for name, got, want in (("df", fr.grad, f64.grad), ("dw", wr.grad, w64.grad), ("dgamma", gr.grad, g64.grad),
                            ("dbeta", br.grad, b64.grad)):
        err = float((got.double() - want).abs().max())
        assert err <= 1e-4 * max(1.0, float(want.abs().max())), (name, err, float(want.abs().max()))


@pytest.mark.parametrize("B,Cin,C,N,M,radius", [(3, 8, 8, 1111, 277, 0.06), (2, 16, 16, 2000, 500, 0.05), (3, 32, 32, 1501, 375, 0.07),
                                                (2, 32, 32, 6000, 1500, 0.35), (3, 64, 64, 1203, 300, 0.06), (2, 128, 128, 640, 160, 0.1)])
def test_gather_backward_on_hubs_and_empty_lists(B, Cin, C, N, M, radius):
    """GroupedConvBN's backward over reverse lists (csrc/csr.hip: the streaming kernel below 64 channels, the point-by-point one
    with lane-distributed records from 64 up) against its float-atomic form, on neighbourhoods as a ball query makes them in a
    sparse cloud: balls with a handful of points are padded with repeats of their first point (lists of 30-200 edges among
    lists of a few) and many points are in no ball at all (empty lists); cloud sizes that no group size divides.  Every
    channel width that selects another kernel instance; with and without the (dp, position) records in list order."""
    from amcontrast3d_amd import ops
    K = 32
    p, idx, dp, f, w, gamma, beta, _ = _case(B, Cin, C, N, M, K, 5 + C + N, True, radius=radius)
    start, edge = ops.group_csr(idx, N)
    deg = np.diff(start.cpu().numpy())
    if radius < 0.2:
        assert deg.max() >= 30 and (deg == 0).mean() > 0.2, (deg.max(), (deg == 0).mean())
    go = torch.randn(B, C, M, K, generator=torch.Generator().manual_seed(4)).to(f.device)
    mom = ops.group_moments(idx, dp, N)
    edge_dp = ops.group_csr_dp(idx, dp, edge)
    grads = []
    for csr in (None, (start, edge), (start, edge, edge_dp)):
        fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
        x1 = ops.GroupedConvBN.apply(fr, dp, idx, mom, wr, gr, br, 1e-5, True, None, csr)
        x1.backward(go)
        grads.append([t.grad.clone() for t in (fr, wr, gr, br)])
    for a, b_ in zip(grads[0], grads[2]):
        assert float((a - b_).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
    assert all(torch.equal(a, b_) for a, b_ in zip(grads[1], grads[2]))  # the two list formats: the same sums in the same order
    # empty lists: exactly zero feature gradient there, in every form
    none = torch.from_numpy((deg == 0).reshape(B, N)).to(f.device)
    assert float(grads[2][0].transpose(1, 2)[none].abs().max()) == 0.0


def test_reverse_lists_of_a_hub_heavy_query_are_ascending():
    """lists longer than the in-place insertion sort handles (a ball query whose centres crowd around a few points): the rank sort
    of csr_order_long_kernel; every list ascending, every position exactly once"""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(11)
    B, N, M, K = 2, 3000, 1200, 32
    idx = torch.randint(0, 12, (B, M, K), generator=g).to(torch.int32)  # twelve hubs: lists of ~3200 positions
    idx[:, ::7, :] = torch.randint(0, N, (B, len(range(0, M, 7)), K), generator=g).to(torch.int32)
    start, edge = ops.group_csr(idx.to("cuda:0"), N)
    s, e, flat = start.cpu().numpy(), edge.cpu().numpy(), idx.reshape(B, -1).numpy()
    P = M * K
    assert s[0] == 0 and s[-1] == B * P
    for b in range(B):
        assert sorted(e[s[b * N]:s[(b + 1) * N]].tolist()) == list(range(P))
    longest = 0
    for gidx in range(B * N):
        lst = e[s[gidx]:s[gidx + 1]]
        if len(lst) > 1:
            assert np.all(np.diff(lst) > 0), gidx
        b, j = divmod(gidx, N)
        assert np.all(flat[b][lst] == j)
        longest = max(longest, len(lst))
    assert longest > 1000


def test_reverse_lists_moments_and_gather_backward():
    """csrc/csr.hip: the reverse adjacency of a ball query (every position exactly once, under its target, ascending), the
    geometry moments derived from it (identical in-degree, dp sums equal to the fixed-point atomics' up to the double
    summation order), and GroupedConvBN's backward as a gather over the lists against its float-atomic form."""
    from amcontrast3d_amd import ops, timing
    B, Cin, C, N, M, K = 3, 32, 64, 1500, 375, 32
    p, idx, dp, f, w, gamma, beta, _ = _case(B, Cin, C, N, M, K, 77, True)
    start, edge = ops.group_csr(idx, N)
    s, e, flat = start.cpu().numpy(), edge.cpu().numpy(), idx.reshape(B, -1).cpu().numpy()
    P = M * K
    assert s[0] == 0 and s[-1] == B * P and np.all(np.diff(s) >= 0)
    for b in range(B):
        seg = e[s[b * N]:s[(b + 1) * N]]
        assert sorted(seg.tolist()) == list(range(P))  # every position of the batch exactly once
    for g in (0, 7, N + 3, 2 * N + 11, B * N - 1):
        b, j = divmod(g, N)
        lst = e[s[g]:s[g + 1]]
        assert np.all(flat[b][lst] == j) and np.all(np.diff(lst) > 0)
        assert len(lst) == int((flat[b] == j).sum())
    m_atomic = ops.group_moments(idx, dp, N).cpu().numpy()
    m_csr = ops.group_moments_csr(idx, dp, N, (start, edge)).cpu().numpy()
    G = B * N
    np.testing.assert_array_equal(m_atomic[128:128 + 4 * G], m_csr[128:128 + 4 * G])  # in-degrees
    off = 128 + 4 * G + 4 * (G & 1)
    da, dc = m_atomic[off:off + 24 * G].view(np.int64), m_csr[off:off + 24 * G].view(np.int64)
    assert np.abs(da - dc).max() <= 64  # fixed point 2^-36: per-term rounding (atomics) vs one rounding of the double sum
    assert np.abs(m_atomic[:72].view(np.int64) - m_csr[:72].view(np.int64)).max() <= 4096
    go = torch.randn(B, C, M, K, generator=torch.Generator().manual_seed(4)).to(f.device)
    grads = []
    for csr in (None, (start, edge)):
        fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
        mom = ops.group_moments(idx, dp, N)
        x1 = ops.GroupedConvBN.apply(fr, dp, idx, mom, wr, gr, br, 1e-5, True, None, csr)
        x1.backward(go)
        grads.append([t.grad.clone() for t in (fr, wr, gr, br)])
    for a, b_ in zip(*grads):
        assert float((a - b_).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
    # the gather form is bit-reproducible
    fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
    x1 = ops.GroupedConvBN.apply(fr, dp, idx, ops.group_moments(idx, dp, N), wr, gr, br, 1e-5, True, None, (start, edge))
    x1.backward(go)
    assert all(torch.equal(a, t.grad) for a, t in zip(grads[1], (fr, wr, gr, br)))
    # with the (dp, position) stream of the edges in list order (ops.group_csr_dp: what the plan adds since round 3) the
    # gather reads one 16-byte record per edge instead of an id and three scattered floats: the same bits
    edge_dp = ops.group_csr_dp(idx, dp, edge)
    ed = edge_dp.cpu().numpy()
    assert np.array_equal(ed[:, 3].copy().view(np.int32), e)
    dpn = dp.reshape(B, 3, P).cpu().numpy()
    for g in (0, 5, P + 9, B * P - 1):
        assert np.array_equal(ed[g, :3], dpn[g // P][:, e[g]])
    fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
    x1 = ops.GroupedConvBN.apply(fr, dp, idx, ops.group_moments(idx, dp, N), wr, gr, br, 1e-5, True, None, (start, edge, edge_dp))
    x1.backward(go)
    assert all(torch.equal(a, t.grad) for a, t in zip(grads[1], (fr, wr, gr, br)))
    # ... and reads a gradient that arrives as position-major rows (the layout SATailActivated writes) in place
    fr, wr, gr, br = (t.clone().requires_grad_(True) for t in (f, w, gamma, beta))
    x1 = ops.GroupedConvBN.apply(fr, dp, idx, ops.group_moments(idx, dp, N), wr, gr, br, 1e-5, True, None, (start, edge))
    go_pm = go.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not go_pm.is_contiguous() and torch.equal(go_pm, go)
    with timing.count_calls() as calls:
        x1.backward(go_pm)
    assert all(torch.equal(a, t.grad) for a, t in zip(grads[1], (fr, wr, gr, br)))
