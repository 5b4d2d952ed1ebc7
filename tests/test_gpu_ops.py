"""HIP kernels (through the C-ABI, via amcontrast3d_amd.ops / compat) against the oracle's C
restatement on identical seeded inputs, against the golden fixture, and -- at benchmark size --
through size-independent properties.  Indices must be bit-exact; fp32 values that involve no
reassociation must be bit-exact too; scatter-add gradients (atomics) get 1e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TAGS = ["room", "lattice", "dup"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from amcontrast3d_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops_fix():
    return load_golden("ops_small")


def clouds(seed, B, N, kind):
    from amcontrast3d_amd import synthetic
    rng = np.random.default_rng(seed)
    if kind == "room":
        return synthetic.make_batch(B, N, first_id=seed)["pos"]
    if kind == "uniform":
        return rng.uniform(0, 2, size=(B, N, 3)).astype(np.float32)
    if kind == "lattice":  # exact distance ties everywhere
        return (rng.integers(0, 6, size=(B, N, 3)) * 0.25).astype(np.float32)
    if kind == "dup":
        base = rng.uniform(0, 1, size=(B, max(N // 3, 1), 3)).astype(np.float32)
        return np.ascontiguousarray(base[:, rng.integers(0, base.shape[1], size=N)])
    raise ValueError(kind)


# ---------------------------------------------------------------------------------- ball query
@pytest.mark.parametrize("tag", TAGS)
def test_ball_query_golden(dev, ops_fix, tag):
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(ops_fix[f"{tag}/xyz"]).to(dev)
    fps = torch.from_numpy(ops_fix[f"{tag}/fps"]).to(dev)
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    got = ops.ball_query(float(ops_fix[f"{tag}/ball_radius"]), 32, xyz, new_xyz)
    np.testing.assert_array_equal(got.cpu().numpy(), ops_fix[f"{tag}/ball"])


@pytest.mark.parametrize("B,N,M,r,ns,kind", [
    (2, 1000, 250, 0.2, 32, "room"), (3, 1031, 517, 0.15, 16, "uniform"), (1, 300, 300, 0.3, 32, "lattice"),
    (2, 64, 7, 0.05, 8, "uniform"), (1, 5000, 1250, 0.1, 32, "room"), (2, 700, 100, 10.0, 32, "dup"),
    (1, 1, 1, 0.5, 4, "uniform"), (2, 2100, 33, 0.25, 70, "uniform"), (1, 3000, 20, 1e-6, 32, "uniform"),
    # n*m >= 2^21: the grid search (cell edge >= radius) instead of the all-pairs scan
    (2, 4096, 1024, 0.3, 32, "lattice"), (1, 6000, 1500, 10.0, 32, "dup"), (2, 3000, 1000, 1e-6, 32, "uniform"),
    (1, 8000, 600, 0.12, 64, "room"), (3, 2500, 900, 0.4, 5, "uniform"), (1, 2200, 1000, 0.25, 70, "uniform")])
def test_ball_query_vs_oracle(dev, B, N, M, r, ns, kind):
    from amcontrast3d_amd import ops
    from oracle import pointops_ref as K
    xyz = torch.from_numpy(clouds(11, B, N, kind))
    q = xyz[:, torch.randperm(N, generator=torch.Generator().manual_seed(1))[:M]].contiguous()
    want = K.ball_query(r, ns, xyz, q)
    got = ops.ball_query(r, ns, xyz.to(dev), q.to(dev))
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())


def test_ball_query_full_size_properties(dev):
    """BASELINE config 2 shape (B=8, 24000 -> 6000, r=0.1, 32): every returned index is inside the
    radius, rows are ascending up to the hit count then padded with the first hit, and the hit
    count equals min(32, #in-radius) -- checked against a blocked torch distance matrix."""
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(clouds(0, 8, 24000, "room")).to(dev)
    q = xyz[:, ::4].contiguous()
    idx = ops.ball_query(0.1, 32, xyz, q).long()
    r2 = torch.tensor(0.1, dtype=torch.float32, device=dev) ** 2
    for b in range(8):
        d = xyz[b][None, :, :] - q[b][:, None, :]
        D = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        inside = D < r2
        cnt = inside.sum(1).clamp(max=32)
        assert bool(torch.gather(inside, 1, idx[b]).all())
        ar = torch.arange(32, device=dev)[None, :]
        live = ar < cnt[:, None]
        # the k-th returned index is the k-th in-radius index
        rank = torch.cumsum(inside, 1) - 1
        kth = torch.full((q.shape[1], 32), -1, device=dev, dtype=torch.long)
        rows, cols = torch.nonzero(inside & (rank < 32), as_tuple=True)
        kth[rows, rank[rows, cols]] = cols
        assert bool((idx[b][live] == kth[live]).all())
        assert bool((idx[b][~live] == idx[b][:, :1].expand(-1, 32)[~live]).all())


# ---------------------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("tag", TAGS)
def test_fps_golden(dev, ops_fix, tag):
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(ops_fix[f"{tag}/xyz"]).to(dev)
    got = ops.furthest_point_sample(xyz, xyz.shape[1] // 4)
    np.testing.assert_array_equal(got.cpu().numpy(), ops_fix[f"{tag}/fps"])


@pytest.mark.parametrize("B,N,M,kind", [
    (2, 1000, 250, "room"), (3, 375, 93, "uniform"), (2, 93, 23, "uniform"), (1, 6000, 1500, "room"),
    (2, 512, 512, "lattice"), (2, 300, 300, "dup"), (1, 40, 10, "dup"), (1, 1, 1, "uniform"), (2, 2, 2, "uniform"),
    (1, 24000, 600, "room"), (1, 4097, 300, "lattice"), (1, 24577, 64, "uniform"), (1, 40000, 50, "uniform"),
    # every points-per-thread bracket of the register kernel, tie-heavy clouds included (slow path)
    (2, 24000, 1500, "dup"), (1, 13000, 500, "room"), (1, 12288, 300, "lattice"), (1, 7000, 700, "dup"),
    (2, 3100, 400, "uniform"), (1, 1600, 1600, "lattice"), (3, 600, 150, "dup"), (1, 65, 65, "uniform"),
    (1, 64, 64, "lattice"), (2, 1537, 200, "room"), (1, 24576, 200, "uniform"),
    # above 24576 points per cloud: the L2-resident pruned kernel (ScanNet-sized batches of the MM configs)
    (2, 64000, 1000, "room"), (1, 30000, 800, "dup"), (1, 50000, 400, "lattice"), (1, 120000, 300, "room"),
    (1, 24577, 24577, "uniform")])
def test_fps_vs_oracle(dev, B, N, M, kind):
    from amcontrast3d_amd import ops
    from oracle import pointops_ref as K
    xyz = torch.from_numpy(clouds(5, B, N, kind))
    want = K.furthest_point_sample(xyz, M)
    got = ops.furthest_point_sample(xyz.to(dev), M)
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())


def test_fps_full_size_properties(dev):
    """B=8, 24000 -> 6000: picks are distinct, start at 0, and each pick maximises the running
    minimum distance to the earlier picks (verified for a strided subset of iterations)."""
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(clouds(0, 8, 24000, "room")).to(dev)
    idx = ops.furthest_point_sample(xyz, 6000).long()
    assert bool((idx[:, 0] == 0).all())
    for b in range(8):
        assert idx[b].unique().numel() == 6000
    b = 3
    mind = torch.full((24000,), 1e10, device=dev)
    for j in range(1, 400):
        c = xyz[b, idx[b, j - 1]]
        d = xyz[b] - c
        mind = torch.minimum(mind, (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        assert float(mind[idx[b, j]]) == float(mind.max())


# ---------------------------------------------------------------------- grouping / gather
def test_grouping_forward_backward(dev, ops_fix):
    from amcontrast3d_amd import ops
    feats = torch.from_numpy(ops_fix["grad/feats"]).to(dev).requires_grad_(True)
    idx = torch.from_numpy(ops_fix["grad/idx"]).to(dev)
    g = torch.from_numpy(ops_fix["grad/g"]).to(dev)
    out = ops.grouping_operation(feats, idx)
    want = torch.stack([feats.detach()[b][:, idx[b].long()] for b in range(feats.shape[0])])
    assert torch.equal(out.detach(), want)
    out.backward(g)
    np.testing.assert_allclose(feats.grad.cpu().numpy(), ops_fix["grad/group_grad"], rtol=1e-5, atol=1e-5)
    # gather_operation = grouping with one sample
    gi = idx[:, :, 0].contiguous()
    got = ops.gather_operation(feats.detach(), gi)
    assert torch.equal(got, torch.gather(feats.detach(), 2, gi.long().unsqueeze(1).expand(-1, feats.shape[1], -1)))


@pytest.mark.parametrize("tag", TAGS)
def test_query_and_group_golden(dev, ops_fix, tag):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models.layers import QueryAndGroup
    xyz = torch.from_numpy(ops_fix[f"{tag}/xyz"]).to(dev)
    fps = torch.from_numpy(ops_fix[f"{tag}/fps"]).to(dev)
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.from_numpy(ops_fix[f"{tag}/feats"]).to(dev)
    dp, fj = QueryAndGroup(float(ops_fix[f"{tag}/ball_radius"]), 32, normalize_dp=True)(new_xyz, xyz, feats)
    np.testing.assert_array_equal(fj.cpu().numpy(), ops_fix[f"{tag}/group_fj"])
    # (x - c) / r : the GPU divides like the CPU reference path (<= 1 ulp otherwise)
    np.testing.assert_allclose(dp.cpu().numpy(), ops_fix[f"{tag}/group_dp"], rtol=2e-7, atol=0)


# ------------------------------------------------------------------------ 3-NN / interpolate
@pytest.mark.parametrize("tag", TAGS)
def test_three_nn_and_interpolation_golden(dev, ops_fix, tag):
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(ops_fix[f"{tag}/xyz"]).to(dev)
    fps = torch.from_numpy(ops_fix[f"{tag}/fps"]).to(dev)
    known = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    dist, idx = ops.three_nn(xyz, known)
    np.testing.assert_array_equal(idx.cpu().numpy(), ops_fix[f"{tag}/three_nn_idx"])
    np.testing.assert_allclose(dist.cpu().numpy(), ops_fix[f"{tag}/three_nn_dist"], rtol=2e-7, atol=0)  # sqrt ulp
    cf = torch.from_numpy(ops_fix[f"{tag}/coarse_feats"]).to(dev)
    got = ops.three_interpolation(xyz, known, cf)
    np.testing.assert_allclose(got.cpu().numpy(), ops_fix[f"{tag}/interp"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,N,M,kind", [(2, 1000, 250, "room"), (1, 517, 3, "uniform"), (2, 300, 2, "uniform"),
                                        (1, 2000, 1031, "lattice"), (2, 6000, 1500, "room"), (1, 50, 1, "dup"),
                                        # n*m >= 2^21: grid search; lattice/dup = equal distances everywhere
                                        (1, 4000, 1000, "lattice"), (2, 5000, 700, "dup"), (1, 3000, 800, "uniform"),
                                        (3, 1500, 1500, "room"),
                                        # the finest FeaturePropagation level of the benchmarked step (per cloud)
                                        (2, 24000, 6000, "room")])
def test_three_nn_vs_oracle(dev, B, N, M, kind):
    from amcontrast3d_amd import _lib, ops
    from oracle import pointops_ref as K
    unknown = torch.from_numpy(clouds(3, B, N, kind))
    known = torch.from_numpy(clouds(4, B, M, kind))
    d2w = torch.empty(B, N, 3)
    iw = torch.empty(B, N, 3, dtype=torch.int32)
    K.three_nn_wrapper(B, N, M, unknown, known, d2w, iw)
    from amcontrast3d_amd import compat
    d2g = torch.empty(B, N, 3, device=dev)
    ig = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
    compat.three_nn_wrapper(B, N, M, unknown.to(dev), known.to(dev), d2g, ig)
    np.testing.assert_array_equal(ig.cpu().numpy(), iw.numpy())
    np.testing.assert_array_equal(d2g.cpu().numpy(), d2w.numpy())  # squared distances: bit-exact (inf included)


def test_three_interpolate_backward(dev):
    from amcontrast3d_amd import ops
    from oracle import pointops_ref as K
    g = torch.Generator().manual_seed(0)
    B, C, M, N = 2, 7, 40, 333
    feats = torch.randn(B, C, M, generator=g)
    idx = torch.randint(0, M, (B, N, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, N, 3, generator=g)
    go = torch.randn(B, C, N, generator=g)
    want_out = torch.empty(B, C, N)
    K.three_interpolate_wrapper(B, C, M, N, feats, idx, w, want_out)
    want_grad = torch.zeros(B, C, M)
    K.three_interpolate_grad_wrapper(B, C, N, M, go, idx, w, want_grad)
    f = feats.to(dev).requires_grad_(True)
    out = ops.three_interpolate(f, idx.to(dev), w.to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want_out.numpy())
    out.backward(go.to(dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), want_grad.numpy(), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------- k-NN
@pytest.mark.parametrize("tag", TAGS)
def test_knn_golden(dev, ops_fix, tag):
    from amcontrast3d_amd import ops
    xyz = ops_fix[f"{tag}/xyz"]
    B, N, _ = xyz.shape
    flat = torch.from_numpy(xyz.reshape(-1, 3).copy()).to(dev)
    for seg, off in (("one", [B * N]), ("per", [N * (b + 1) for b in range(B)])):
        o = torch.tensor(off, dtype=torch.int32, device=dev)
        idx, dist = ops.knnquery(24, flat, flat, o, o)
        np.testing.assert_array_equal(idx.cpu().numpy(), ops_fix[f"{tag}/knn24_{seg}_idx"])
        np.testing.assert_allclose(dist.cpu().numpy(), ops_fix[f"{tag}/knn24_{seg}_dist"], rtol=2e-7, atol=0)
    fps = ops_fix[f"{tag}/fps"]
    q = torch.from_numpy(np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1).reshape(-1, 3).copy()).to(dev)
    o = torch.tensor([B * N], dtype=torch.int32, device=dev)
    qo = torch.tensor([q.shape[0]], dtype=torch.int32, device=dev)
    idx, dist = ops.knnquery(64, flat, q, o, qo)
    np.testing.assert_array_equal(idx.cpu().numpy(), ops_fix[f"{tag}/knn64_idx"])


@pytest.mark.parametrize("n,m,k,segs,kind", [
    (2000, 2000, 24, 1, "room"), (3000, 700, 16, 3, "uniform"), (1500, 1500, 24, 2, "lattice"),
    (900, 900, 24, 1, "dup"), (10, 10, 24, 1, "uniform"), (5000, 300, 64, 1, "room"), (4000, 100, 100, 2, "uniform"),
    (257, 1029, 4, 1, "uniform"), (1, 3, 1, 1, "uniform"),
    # above the brute-force threshold (n*m >= 2^22): the grid path, incl. its tie fallback
    (6000, 6000, 24, 1, "lattice"), (6000, 6000, 24, 2, "dup"), (20000, 3000, 64, 1, "room"),
    (8000, 8000, 16, 3, "uniform"), (30000, 5000, 4, 1, "room"), (12000, 12000, 24, 1, "room"),
    (5000, 5000, 1, 1, "uniform"), (70000, 64, 24, 1, "uniform")])
def test_knn_vs_oracle(dev, n, m, k, segs, kind):
    """including ragged segments, fewer points than k (placeholders), k = 100 (the reference's cap)
    and tie-heavy clouds: indices bit-exact, i.e. the heap's tie order is reproduced"""
    from amcontrast3d_amd import compat
    from oracle import pointops_ref as K
    rng = np.random.default_rng(n + m)
    xyz = torch.from_numpy(clouds(9, 1, n, kind)[0])
    if m == n:
        q = xyz.clone()
    else:
        q = torch.from_numpy(clouds(10, 1, m, kind)[0])
    cut = np.sort(rng.choice(np.arange(1, n), size=segs - 1, replace=False)) if segs > 1 else np.array([], dtype=int)
    off = torch.tensor(list(cut) + [n], dtype=torch.int32)
    qcut = np.sort(rng.choice(np.arange(1, m), size=segs - 1, replace=False)) if segs > 1 else np.array([], dtype=int)
    qoff = torch.tensor(list(qcut) + [m], dtype=torch.int32)
    iw = torch.zeros(m, k, dtype=torch.int32)
    dw = torch.zeros(m, k)
    K.knnquery_cuda(m, k, xyz, q, off, qoff, iw, dw)
    ig = torch.zeros(m, k, dtype=torch.int32, device=dev)
    dg = torch.zeros(m, k, device=dev)
    compat.knnquery_cuda(m, k, xyz.to(dev), q.to(dev), off.to(dev), qoff.to(dev), ig, dg)
    np.testing.assert_array_equal(dg.cpu().numpy(), dw.numpy())
    np.testing.assert_array_equal(ig.cpu().numpy(), iw.numpy())


def test_knn_full_size_properties(dev):
    """Loss stage 1 shape (48000 points of 8 clouds in one segment, k=24): first neighbour is the point
    itself at distance 0, rows ascend, and the k-th distance equals the k-th smallest of a blocked
    brute-force distance matrix."""
    from amcontrast3d_amd import ops
    p = torch.from_numpy(clouds(0, 8, 6000, "room").reshape(-1, 3)).to(dev)
    o = torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
    idx, dist = ops.knnquery(24, p, p, o, o)
    assert bool((dist[:, 0] == 0).all())
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    sel = torch.arange(0, p.shape[0], 37, device=dev)
    d = p[sel][:, None, :] - p[None, :, :]
    D = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    want = torch.sqrt(torch.topk(D, 24, dim=1, largest=False).values)
    assert torch.equal(dist[sel], want)
    assert torch.equal(torch.sqrt(torch.gather(D, 1, idx[sel].long())), want)


# ------------------------------------------------------------------------------------- errors
def test_cpu_tensors_are_rejected_loudly(dev):
    from amcontrast3d_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.furthest_point_sample(torch.rand(1, 10, 3), 2)
    with pytest.raises(RuntimeError, match="nsample"):
        p = torch.rand(10, 3, device=dev)
        o = torch.tensor([10], dtype=torch.int32, device=dev)
        ops.knnquery(101, p, p, o, o)


def test_knn_grid_reuse_gives_the_same_lists(dev):
    """Searches over one support cloud inside `knn_grid_reuse()` share the first call's cell grid (calibrated for
    that call's k): results must equal independent calls, also when the first call was too small to build a grid."""
    from amcontrast3d_amd import ops
    xyz = torch.from_numpy(clouds(21, 1, 30000, "room")[0]).to(dev)
    o = torch.tensor([xyz.shape[0]], dtype=torch.int32, device=dev)
    queries = [(24, xyz, o), (4, xyz[::4].contiguous(), None), (64, xyz[::64].contiguous(), None),
               (16, xyz[:50].contiguous(), None)]  # the last one: 50 x 30000 pairs -> all-pairs kernel
    def run(k, q, qo):
        qo = qo if qo is not None else torch.tensor([q.shape[0]], dtype=torch.int32, device=dev)
        idx, dist = ops.knnquery(k, xyz, q, o, qo)
        return idx.clone(), dist.clone()
    alone = [run(*a) for a in queries]
    for order in (queries, queries[::-1]):  # reversed: the small all-pairs call comes first and must not be "reused"
        with ops.knn_grid_reuse():
            shared = [run(*a) for a in order]
        if order is not queries:
            shared = shared[::-1]
        for (i0, d0), (i1, d1) in zip(alone, shared):
            assert torch.equal(i0, i1) and torch.equal(d0, d1)
