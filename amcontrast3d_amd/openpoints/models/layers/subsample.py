"""Sub-sampling (models/layers/subsample.py:70-157)."""
import torch

from amcontrast3d_amd.ops import furthest_point_sample, gather_operation  # noqa: F401


def random_sample(xyz, npoint):
    B, N, _ = xyz.shape
    return torch.randint(0, N, (B, npoint), device=xyz.device)


def fps(data, number):
    """data (B,N,C>=3) -> the `number` furthest-point-sampled rows"""
    idx = furthest_point_sample(data[:, :, :3].contiguous(), number)
    return torch.gather(data, 1, idx.unsqueeze(-1).long().expand(-1, -1, data.shape[-1]))
