"""Device input pipeline (csrc/voxel.hip, amcontrast3d_amd/input_pipeline.py) against what the reference's own
voxelize / crop_pc returned for the same cloud and the same random draws (tests/golden/input_room.npz), against the
oracle on a second seeded cloud, and by size-independent properties at loader scale (a 1.2 M-point raw room)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _sets(idx_sort, count):
    start = np.concatenate([[0], np.cumsum(count)])
    return [frozenset(idx_sort[start[v]:start[v + 1]].tolist()) for v in range(len(count))]


def test_voxelize_and_crop_match_the_reference_run():
    from amcontrast3d_amd import input_pipeline as ip
    g = load_golden("input_room")
    dev = torch.device("cuda:0")
    coord = torch.from_numpy(g["coord"]).to(dev)
    voxel = float(g["voxel"])
    key, idx_sort, voxel_idx, start, count = ip._voxel_tables(coord, voxel)
    np.testing.assert_array_equal(key.cpu().numpy().view(np.uint64), g["key"])          # FNV-1a hash, bit for bit
    np.testing.assert_array_equal(count.cpu().numpy(), g["count"])
    np.testing.assert_array_equal(voxel_idx.cpu().numpy(), g["voxel_idx"])
    isort = idx_sort.cpu().numpy()
    np.testing.assert_array_equal(g["key"][isort], g["key"][g["idx_sort"]])
    assert _sets(isort, g["count"]) == _sets(g["idx_sort"], g["count"])
    for v in (0, 1, len(g["count"]) // 2):  # stable: ascending point index inside a voxel
        s, c = int(start[v]), int(count[v])
        assert np.all(np.diff(isort[s:s + c]) > 0)
    # val mode through the mirror function
    a, b, c = ip.voxelize(coord, voxel, mode=1)
    assert a.dtype == torch.int64 and torch.equal(c.cpu(), torch.from_numpy(g["count"]))
    # train mode with the reference's own draw: one point of the same voxel each
    pick = ip.voxelize(coord, voxel, mode=0, rnd=torch.from_numpy(g["rnd"])).cpu().numpy()
    np.testing.assert_array_equal(g["key"][pick], g["key"][g["idx_unique"]])
    # crop of the voxelised cloud around the centre point (validation split)
    cv = torch.from_numpy(g["coord"][g["idx_unique"]]).to(dev)
    d2, crop_idx = ip.crop_nearest(cv, len(cv) // 2, int(g["voxel_max"]))
    np.testing.assert_array_equal(d2.cpu().numpy(), g["d2"])
    ci = crop_idx.cpu().numpy()
    np.testing.assert_array_equal(g["d2"][ci], g["d2"][g["crop_idx"]])
    assert set(ci.tolist()) == set(g["crop_idx"].tolist()) or np.sum(g["d2"] == g["d2"][g["crop_idx"][-1]]) > 1
    cc, ff, ll = ip.crop_pc(cv, torch.from_numpy(g["feat"][g["idx_unique"]]).to(dev),
                            torch.from_numpy(g["label"][g["idx_unique"]]).to(dev), "val", voxel, int(g["voxel_max"]),
                            downsample=False, shuffle=False)
    if np.array_equal(ci, g["crop_idx"]):
        np.testing.assert_array_equal(cc.cpu().numpy(), g["crop_coord"])
        np.testing.assert_array_equal(ff.cpu().numpy(), g["crop_feat"])
        np.testing.assert_array_equal(ll.cpu().numpy(), g["crop_label"])


def test_against_oracle_and_properties_at_loader_scale():
    from amcontrast3d_amd import input_pipeline as ip
    from oracle import input_ref
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    # a raw room: 1.2 M points on a few surfaces, ~10 per 4 cm voxel
    n = 1_200_000
    coord = np.stack([rng.uniform(0, 6, n), rng.uniform(0, 5, n), rng.choice([0.0, 1.0, 2.9], n) + rng.uniform(0, 0.05, n)], 1)
    coord = (coord - coord.min(0)).astype(np.float32)
    want = input_ref.voxelize(coord, 0.04, mode=1)
    g = torch.from_numpy(coord).to(dev)
    idx_sort, voxel_idx, count = ip.voxelize(g, 0.04, mode=1)
    np.testing.assert_array_equal(idx_sort.cpu().numpy(), want[0])   # stable order on both sides: identical
    np.testing.assert_array_equal(voxel_idx.cpu().numpy(), want[1])
    np.testing.assert_array_equal(count.cpu().numpy(), want[2])
    assert int(count.sum()) == n and sorted(idx_sort.cpu().tolist()) == list(range(n))
    gen = torch.Generator(device=dev).manual_seed(1)
    cc, ff, ll = ip.crop_pc(g, g.clone(), torch.arange(n, device=dev), "train", 0.04, 24000, generator=gen)
    assert cc.shape == (24000, 3) and float(cc.min()) == 0.0 and ll.dtype == torch.int64
    # one point per voxel: the cropped points' cells are pairwise distinct, and features travelled with their points
    cells = torch.floor((g[ll] - g.min(0).values).double() / 0.04).long()
    assert len(torch.unique(cells, dim=0)) == 24000
    assert torch.equal(ff, g[ll])
    # N < voxel_max: variable clouds are left as they are, fixed-size ones are padded by repetition
    small = g[:5000]
    c2, _, l2 = ip.crop_pc(small, None, torch.arange(5000, device=dev), "train", 0.04, 6000, downsample=False, variable=False,
                           generator=gen)
    assert c2.shape[0] == 6000 and len(torch.unique(l2)) == 5000
    c3, _, l3 = ip.crop_pc(small, None, torch.arange(5000, device=dev), "val", 0.04, 6000, downsample=False, shuffle=False)
    assert torch.equal(l3, torch.arange(5000, device=dev))
