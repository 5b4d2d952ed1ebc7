"""The loader's per-cloud preprocessing on the MI355X.

Same names, arguments and results as the reference's numpy functions
    voxelize(coord, voxel_size, hash_type, mode)      openpoints/dataset/data_util.py:127-141
    crop_pc(coord, feat, label, split, voxel_size, voxel_max, downsample, variable, shuffle)   :146-174
on torch GPU tensors, running csrc/voxel.hip: FNV-1a cell hash, stable radix sort, run-length voxel ids / counts, nearest-
`voxel_max` crop, min-corner shift.  The reference runs them in 6 numpy loader workers per GPU
(cfgs/s3dis/default.yaml:30-31), which cannot feed a step of ~10 ms; S3DIS.__getitem__ (dataset/s3dis/s3dis.py:122-144)
calls crop_pc once per cloud, and that call is what these replace.

Randomness: the reference draws from numpy's global RandomState (np.random.randint / choice / permutation).  Here the
draws come from a torch.Generator on the device, or are passed in (`rnd`, `init_idx`, `perm`) -- the tests pass the very
numbers the reference drew.  numpy's argsort is not stable, so WHICH point of a voxel the reference's mode-0 pick lands on
(and the order of equidistant points in the crop) is unspecified by the reference itself; the stable order is used here.
"""
import ctypes

import torch

from . import _lib
from .ops import _need_dtype, _need_gpu, _ptr, _stream


def _voxel_tables(coord, voxel_size):
    _need_gpu(coord)
    _need_dtype(torch.float32, coord=coord)
    coord = coord.contiguous()
    n = coord.shape[0]
    dev = coord.device
    lib = _lib.load()
    key = torch.empty(n, dtype=torch.int64, device=dev)  # uint64 bit patterns
    idx_sort = torch.empty(n, dtype=torch.int32, device=dev)
    voxel_idx = torch.empty(n, dtype=torch.int32, device=dev)
    start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    count = torch.empty(n, dtype=torch.int32, device=dev)
    nvox = torch.empty(1, dtype=torch.int32, device=dev)
    wb = int(lib.amc3d_voxelize_workspace_bytes(n))
    work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.amc3d_voxelize(n, _ptr(coord), ctypes.c_double(float(voxel_size)), _ptr(key), _ptr(idx_sort),
                                      _ptr(voxel_idx), _ptr(start), _ptr(count), _ptr(nvox), _ptr(work), wb, _stream(coord)),
                   "voxelize")
    nv = int(nvox.item())  # the number of voxels sizes what follows: one read-back per cloud
    return key, idx_sort, voxel_idx, start[:nv + 1], count[:nv]


def voxelize(coord, voxel_size=0.05, hash_type='fnv', mode=0, rnd=None, generator=None):
    """coord (n,3) fp32 on the GPU, already shifted to its min corner.
    mode 0 (train): idx_unique (nvox) int64 -- one point per voxel, the rnd[v] % count[v]-th of the voxel, rnd =
    randint(0, count.max(), nvox) drawn on the device (or given);  mode 1 (val): (idx_sort, voxel_idx, count) int64."""
    if hash_type != 'fnv':
        raise NotImplementedError("hash_type 'ravel': the loaders of the AMContrast3D configs use the default 'fnv'")
    key, idx_sort, voxel_idx, start, count = _voxel_tables(coord, voxel_size)
    if mode != 0:
        return idx_sort.long(), voxel_idx.long(), count.long()
    nv = count.shape[0]
    if rnd is None:
        rnd = torch.randint(0, int(count.max().item()), (nv,), device=coord.device, generator=generator, dtype=torch.int32)
    rnd = rnd.to(device=coord.device, dtype=torch.int32).contiguous()
    out = torch.empty(nv, dtype=torch.int32, device=coord.device)
    with torch.cuda.device(coord.device):
        _lib.check(_lib.load().amc3d_voxel_select(nv, _ptr(start), _ptr(count), _ptr(idx_sort), _ptr(rnd), _ptr(out),
                                                  _stream(coord)), "voxel_select")
    return out.long()


def crop_nearest(coord, init_idx, keep):
    """the `keep` points nearest to coord[init_idx], ascending distance -> (d2 (n) fp32, crop_idx (keep) int64)"""
    _need_gpu(coord)
    _need_dtype(torch.float32, coord=coord)
    coord = coord.contiguous()
    n = coord.shape[0]
    dev = coord.device
    lib = _lib.load()
    d2 = torch.empty(n, dtype=torch.float32, device=dev)
    idx = torch.empty(int(keep), dtype=torch.int32, device=dev)
    wb = int(lib.amc3d_crop_nearest_workspace_bytes(n))
    work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.amc3d_crop_nearest(n, _ptr(coord), int(init_idx), int(keep), _ptr(d2), _ptr(idx), _ptr(work), wb,
                                          _stream(coord)), "crop_nearest")
    return d2, idx.long()


def crop_pc(coord, feat, label, split='train', voxel_size=0.04, voxel_max=None, downsample=True, variable=True,
            shuffle=True, generator=None, rnd=None, init_idx=None, perm=None):
    """data_util.py:146-174 on GPU tensors: coord (n,3) fp32, feat (n,c) or None, label (n[,1]) or None ->
    (coord fp32 shifted to its min corner, feat fp32, label int64), all on the GPU."""
    if voxel_size and downsample:
        coord = coord - coord.min(0).values
        uniq = voxelize(coord, voxel_size, rnd=rnd, generator=generator)
        coord = coord[uniq]
        feat = feat[uniq] if feat is not None else None
        label = label[uniq] if label is not None else None
    if voxel_max is not None:
        crop_idx = None
        N = len(label)
        dev = coord.device
        if N >= voxel_max:
            if init_idx is None:
                init_idx = int(torch.randint(N, (1,), generator=generator, device=dev).item()) if 'train' in split else N // 2
            crop_idx = crop_nearest(coord.contiguous(), init_idx, voxel_max)[1]
        elif not variable:  # fill up by repetition (batched data of a fixed size)
            pad = torch.randint(N, (voxel_max - N,), generator=generator, device=dev)
            crop_idx = torch.cat([torch.arange(N, device=dev), pad])
        if crop_idx is None:
            crop_idx = torch.arange(coord.shape[0], device=dev)
        if shuffle:
            if perm is None:
                perm = torch.randperm(len(crop_idx), generator=generator, device=dev)
            crop_idx = crop_idx[perm]
        coord = coord[crop_idx]
        feat = feat[crop_idx] if feat is not None else None
        label = label[crop_idx] if label is not None else None
    coord = coord - coord.min(0).values
    return coord.float(), feat.float() if feat is not None else None, label.long() if label is not None else None
