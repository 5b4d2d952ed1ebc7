"""ORACLE -- test infrastructure only.  numpy restatement of the loader's voxelisation and crop
(openpoints/dataset/data_util.py:92-174), pinned by tests/golden/input_room.npz, which oracle/gen_golden.py records
from the reference's own fnv_hash_vec / voxelize / crop_pc.  Only tests/ import this.

One deliberate difference: the sorts are STABLE.  The reference calls np.argsort with its default (unstable) kind, so
the order of the points inside one voxel, and of two points at exactly the same distance from the crop centre, is an
accident of numpy's sort; every quantity that does not depend on that accident is identical."""
import numpy as np


def fnv_hash_vec(arr):
    """FNV64-1A over the columns of a non-negative (n, d) array (data_util.py:92-105)"""
    arr = arr.astype(np.uint64)
    h = np.full(arr.shape[0], 14695981039346656037, dtype=np.uint64)
    for j in range(arr.shape[1]):
        h *= np.uint64(1099511628211)
        h ^= arr[:, j]
    return h


def voxelize(coord, voxel_size=0.05, mode=0, rnd=None):
    """data_util.py:127-141 -> mode 1: (idx_sort, voxel_idx, count); mode 0: idx_unique for the caller's draw `rnd`
    (= np.random.randint(0, count.max(), count.size) in the reference)"""
    key = fnv_hash_vec(np.floor(coord / np.array(voxel_size)))
    idx_sort = np.argsort(key, kind="stable")
    _, voxel_idx, count = np.unique(key[idx_sort], return_counts=True, return_inverse=True)
    if mode != 0:
        return idx_sort, voxel_idx, count
    start = np.cumsum(np.insert(count, 0, 0)[0:-1])
    return idx_sort[start + rnd % count]


def crop_nearest(coord, init_idx, voxel_max):
    """data_util.py:157-160 -> (d2 float32, crop_idx)"""
    d2 = np.sum(np.square(coord - coord[init_idx]), 1)
    return d2, np.argsort(d2, kind="stable")[:voxel_max]
