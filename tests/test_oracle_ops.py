"""The C restatement of the reference's native kernels (oracle/pointops_ref.c) against
(a) independent brute-force numpy formulations of what each kernel is documented to compute and
(b) the fixture recorded through the reference's own Python wrappers (tests/golden/ops_small.npz).
CPU only.  The reference holds no tests or golden vectors for these kernels."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pointops_ref as K

TAGS = ["room", "lattice", "dup"]


def d2(a, b):
    """fp32, evaluated as the kernels do: ((dx*dx + dy*dy) + dz*dz)"""
    d = (a[:, None, :] - b[None, :, :]).astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


@pytest.fixture(scope="module")
def ops():
    return load_golden("ops_small")


@pytest.mark.parametrize("tag", TAGS)
def test_ball_query_is_first_k_in_radius_by_index(ops, tag):
    xyz = ops[f"{tag}/xyz"]
    r = float(ops[f"{tag}/ball_radius"])
    fps = ops[f"{tag}/fps"]
    got = K.ball_query(r, 32, torch.from_numpy(xyz),
                       torch.from_numpy(np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1).copy())).numpy()
    np.testing.assert_array_equal(got, ops[f"{tag}/ball"])
    r2 = np.float32(r) * np.float32(r)
    for b in range(xyz.shape[0]):
        D = d2(xyz[b][fps[b]], xyz[b])
        for q in range(D.shape[0]):
            hits = np.nonzero(D[q] < r2)[0]
            want = np.zeros(32, dtype=np.int32)
            if len(hits):
                want[:] = hits[0]
                want[:min(32, len(hits))] = hits[:32]
            np.testing.assert_array_equal(got[b, q], want)


@pytest.mark.parametrize("tag", TAGS)
def test_fps_matches_fixture_and_sequential_argmax(ops, tag):
    xyz = ops[f"{tag}/xyz"]
    got = K.furthest_point_sample(torch.from_numpy(xyz), xyz.shape[1] // 4).numpy()
    np.testing.assert_array_equal(got, ops[f"{tag}/fps"])
    assert (got[:, 0] == 0).all()
    if tag == "room":  # no exact ties: any correct FPS gives the same picks
        for b in range(xyz.shape[0]):
            mind = np.full(xyz.shape[1], 1e10, dtype=np.float32)
            cur, picks = 0, [0]
            for _ in range(got.shape[1] - 1):
                mind = np.minimum(mind, d2(xyz[b][cur:cur + 1], xyz[b])[0])
                cur = int(np.argmax(mind))
                picks.append(cur)
            np.testing.assert_array_equal(got[b], picks)


def test_fps_tie_rule_follows_the_reference_block_tree():
    # all points identical except the first: every running minimum ties, so the pick is decided by
    # the reference's block-strided scan + shared-memory tree (sampling_gpu.cu:93-98,150-211):
    # the winner is the candidate whose (bit-reversed thread id, pass) is smallest
    n = 40
    xyz = np.ones((1, n, 3), dtype=np.float32)
    xyz[0, 0] = 0
    got = K.furthest_point_sample(torch.from_numpy(xyz), 3).numpy()[0]
    rb = K.fps_block_size(n)  # 32
    assert rb == 32
    cand = list(range(1, n))
    key = lambda k: (int(format(k % rb, "05b")[::-1], 2), k // rb)
    assert got[1] == min(cand, key=key)
    assert [K.fps_block_size(v) for v in (1, 2, 3, 93, 375, 1024, 1500, 24000)] == [1, 2, 2, 64, 256, 1024, 1024, 1024]


@pytest.mark.parametrize("tag", TAGS)
def test_three_nn_earlier_index_wins_ties(ops, tag):
    xyz = ops[f"{tag}/xyz"]
    fps = ops[f"{tag}/fps"]
    known = np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1).copy()
    dist, idx = K.three_nn(torch.from_numpy(xyz), torch.from_numpy(known))
    np.testing.assert_array_equal(idx.numpy(), ops[f"{tag}/three_nn_idx"])
    np.testing.assert_array_equal(dist.numpy(), ops[f"{tag}/three_nn_dist"])
    for b in range(xyz.shape[0]):
        D = d2(xyz[b], known[b])
        order = np.argsort(D, axis=1, kind="stable")[:, :3]  # stable: equal distances keep index order
        np.testing.assert_array_equal(idx.numpy()[b], order)
        np.testing.assert_array_equal(dist.numpy()[b], torch.sqrt(torch.from_numpy(np.take_along_axis(D, order, 1))).numpy())


@pytest.mark.parametrize("tag", TAGS)
def test_knn_distances_and_fixture(ops, tag):
    xyz = ops[f"{tag}/xyz"]
    B, N, _ = xyz.shape
    flat = torch.from_numpy(xyz.reshape(-1, 3).copy())
    for seg, off in (("one", [B * N]), ("per", [N * (b + 1) for b in range(B)])):
        o = torch.tensor(off, dtype=torch.int32)
        idx, dist = K.knnquery(24, flat, flat, o, o)
        np.testing.assert_array_equal(idx.numpy(), ops[f"{tag}/knn24_{seg}_idx"])
        np.testing.assert_array_equal(dist.numpy(), ops[f"{tag}/knn24_{seg}_dist"])
        # distances: the 24 smallest of the segment, ascending -- independent of tie order
        D = d2(flat.numpy(), flat.numpy())
        if seg == "per":
            segid = np.arange(B * N) // N
            D = np.where(segid[:, None] == segid[None, :], D, np.inf)
        want = torch.sqrt(torch.from_numpy(np.sort(D, axis=1)[:, :24].copy())).numpy()
        np.testing.assert_array_equal(dist.numpy(), want)
        # indices point at points with exactly those distances
        np.testing.assert_array_equal(torch.sqrt(torch.from_numpy(np.take_along_axis(D, idx.numpy().astype(np.int64), 1))).numpy(), want)
        if tag == "room":  # tie-free: unique answer
            np.testing.assert_array_equal(idx.numpy(), np.argsort(D, axis=1, kind="stable")[:, :24])


def test_knn_fewer_points_than_k_keeps_placeholders():
    p = torch.rand(5, 3)
    o = torch.tensor([5], dtype=torch.int32)
    idx, dist = K.knnquery(8, p, p, o, o)
    assert (idx[:, 5:] == 0).all() and np.allclose(dist[:, 5:].numpy(), np.sqrt(np.float32(1e10)))


def test_grouping_and_interpolation_against_numpy(ops):
    xyz, feats, idx = ops["room/xyz"], ops["room/feats"], ops["room/ball"]
    B, C, N = feats.shape
    out = torch.empty(B, C, idx.shape[1], idx.shape[2])
    K.group_points_wrapper(B, C, N, idx.shape[1], idx.shape[2], torch.from_numpy(feats), torch.from_numpy(idx), out)
    want = np.stack([feats[b][:, idx[b]] for b in range(B)])
    np.testing.assert_array_equal(out.numpy(), want)
    np.testing.assert_array_equal(out.numpy(), ops["room/group_fj"])
    # gradient: scatter-add
    g, gi, gf = ops["grad/g"], ops["grad/idx"], ops["grad/feats"]
    acc = np.zeros(gf.shape, dtype=np.float64)
    for b in range(g.shape[0]):
        for c in range(g.shape[1]):
            np.add.at(acc[b, c], gi[b].ravel(), g[b, c].ravel().astype(np.float64))
    np.testing.assert_allclose(ops["grad/group_grad"], acc, rtol=1e-5, atol=1e-5)
    # interpolation weights sum to one and reproduce constants
    fps = ops["room/fps"]
    known = np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1).copy()
    from oracle import model_ref
    ones = torch.ones(B, 2, known.shape[1])
    np.testing.assert_allclose(model_ref.three_interpolation(torch.from_numpy(xyz), torch.from_numpy(known), ones).numpy(),
                               1.0, rtol=0, atol=1e-6)
    got = model_ref.three_interpolation(torch.from_numpy(xyz), torch.from_numpy(known),
                                        torch.from_numpy(ops["room/coarse_feats"])).numpy()
    np.testing.assert_allclose(got, ops["room/interp"], rtol=1e-6, atol=1e-6)


def test_ambiguity_restatement_matches_reference_loop(ops):
    from oracle import model_ref
    p, lab, nidx = torch.from_numpy(ops["amb/p"]), torch.from_numpy(ops["amb/label"]), torch.from_numpy(ops["amb/nidx"])
    posmask = lab[:, None] == lab[nidx.long()]
    a = model_ref.ambiguity(p, posmask, nidx, 0.04)
    np.testing.assert_allclose(a.numpy(), ops["amb/a"], rtol=0, atol=1e-5)
    assert 0.05 < float(((a > 0) & (a < 1)).float().mean()) < 0.6  # the case really has boundary points
