"""Boundary / inner split of the evaluation loops (drop-in for openpoints/AMContrast3D/metrics.py:160-184).

`posmask_searching` marks, for every point, which of its nsample-1 nearest neighbours carry its own label; the
evaluation loops call a point "boundary" when 0 < #same < nsample (examples/segmentation/main_AA.py:470-476).
The reference builds one-hot labels, gathers an (m, nsample-1, classes) tensor and arg-maxes both sides; comparing
the integer labels directly (ops.posmask_from_labels, csrc/loss.hip) is the same predicate.
"""
import torch

from openpoints.cpp.pointops.functions import pointops


def posmask_searching(xyz, target, nsample, num_classes, ignore_index):
    """xyz (m,3) fp32, target (m) int64 -> posmask (m, nsample-1) bool, neighbor_idx (m, nsample-1) int32.
    ignore_index (ScanNet: -100) counts as one extra class, so unlabeled neighbours match unlabeled anchors."""
    from amcontrast3d_amd import ops
    target = target.reshape(-1)
    if ignore_index is not None:
        target = torch.where(target == ignore_index, num_classes, target)
    xyz = xyz.contiguous()
    o = torch.tensor([xyz.shape[0]], dtype=torch.int32, device=xyz.device)
    neighbor_idx, _ = pointops.knnquery(nsample, xyz, xyz, o, o)
    neighbor_idx = neighbor_idx[..., 1:].contiguous()  # drop the self match
    posmask = ops.posmask_from_labels(target.to(torch.int32).contiguous(), neighbor_idx)
    return posmask, neighbor_idx
