"""The oracle's restatement of the loader's voxelize / crop (oracle/input_ref.py) against what the reference's own
functions returned (tests/golden/input_room.npz, recorded by oracle/gen_golden.py input)."""
import numpy as np

from conftest import load_golden


def _per_voxel_sets(idx_sort, count):
    start = np.concatenate([[0], np.cumsum(count)])
    return [frozenset(idx_sort[start[v]:start[v + 1]].tolist()) for v in range(len(count))]


def test_voxelize_matches_reference():
    from oracle import input_ref
    g = load_golden("input_room")
    coord, voxel = g["coord"], float(g["voxel"])
    np.testing.assert_array_equal(input_ref.fnv_hash_vec(np.floor(coord / np.array(voxel))), g["key"])
    idx_sort, voxel_idx, count = input_ref.voxelize(coord, voxel, mode=1)
    np.testing.assert_array_equal(count, g["count"])
    np.testing.assert_array_equal(voxel_idx, g["voxel_idx"])
    np.testing.assert_array_equal(g["key"][idx_sort], g["key"][g["idx_sort"]])  # the same sorted key sequence
    assert _per_voxel_sets(idx_sort, count) == _per_voxel_sets(g["idx_sort"], g["count"])  # the same points per voxel
    pick = input_ref.voxelize(coord, voxel, mode=0, rnd=g["rnd"])
    np.testing.assert_array_equal(g["key"][pick], g["key"][g["idx_unique"]])  # one point of the right voxel each


def test_crop_matches_reference():
    from oracle import input_ref
    g = load_golden("input_room")
    cv = g["coord"][g["idx_unique"]]
    d2, crop_idx = input_ref.crop_nearest(cv, len(cv) // 2, int(g["voxel_max"]))
    np.testing.assert_array_equal(d2, g["d2"])
    boundary = g["d2"][g["crop_idx"][-1]]
    # identical except possibly among points at exactly the boundary distance / equal distances (unstable sort)
    assert set(crop_idx[d2[crop_idx] < boundary].tolist()) == set(g["crop_idx"][g["d2"][g["crop_idx"]] < boundary].tolist())
    np.testing.assert_array_equal(d2[crop_idx], g["d2"][g["crop_idx"]])
    out = (cv[crop_idx] - cv[crop_idx].min(0)).astype(np.float32)
    if np.array_equal(crop_idx, g["crop_idx"]):
        np.testing.assert_array_equal(out, g["crop_coord"])
