"""Ambiguity Prediction Module of AMContrast3D++: per-point MLP regressors from [position ; feature] to a scalar
ambiguity in (0, 1), one per encoder resolution.

Drop-in for openpoints/AMContrast3D/APM/concatenation.py:9-196 (``APM_pf_ConCate``): same constructor keywords,
same module names and nn.Sequential positions, hence the same state-dict keys
(``layer_{s}.{0,4,8,12,16,20}`` Linear, ``.{2,6,10,14,18,21}`` BatchNorm1d, ``map_{s}.0`` Linear), so the
authors' checkpoints load.  The four towers of the reference are written out by hand there; here they are built
by one loop.
"""
from typing import List

import torch
import torch.nn as nn

from openpoints.models.build import MODELS


def _tower(cin: int, channel: List[int], dropout: List[float]) -> nn.Sequential:
    """[Linear, Dropout, BatchNorm1d, Sigmoid] x len(channel), then Linear -> 1, BatchNorm1d, Sigmoid"""
    mods, prev = [], cin
    for width, drop in zip(channel, dropout):
        mods += [nn.Linear(prev, width), nn.Dropout(drop), nn.BatchNorm1d(width), nn.Sigmoid()]
        prev = width
    mods += [nn.Linear(prev, 1), nn.BatchNorm1d(1), nn.Sigmoid()]
    return nn.Sequential(*mods)


@MODELS.register_module()
class APM_pf_ConCate(nn.Module):
    def __init__(self, feature_dim: List[int] = [64, 128, 256, 512], linear_mapping: bool = True,
                 cross_attention: bool = False, feat_concate: bool = True, channel: List[int] = [32, 16, 8, 4, 2],
                 dropout: List[float] = [0, 0, 0, 0, 0], nsample_k: int = 12, threshold: float = 0.7,
                 threshold_max: float = 1.0, gamma: float = 0.5, fusion: str = 'MIN', att_dim: int = 3):
        super().__init__()
        assert len(feature_dim) == 4 and len(channel) == 5 and len(dropout) == 5
        self.dim = list(feature_dim)
        self.map = linear_mapping
        self.drop_rate = list(dropout)
        for s, d in enumerate(self.dim):  # <N x (3+D)> -> <N x 1>
            setattr(self, f'layer_{s}', _tower(3 + d, channel, dropout))
        if self.map:
            for s, d in enumerate(self.dim):  # <N x 1> -> <N x D>
                setattr(self, f'map_{s}', nn.Sequential(nn.Linear(1, d), nn.Sigmoid()))

    def _tower_channel_major(self, tower, p, f):
        """The tower on the tensors as the backbone holds them -- x = [p^T ; f] (B, 3+D, n), every Linear a 1x1
        convolution, every BatchNorm1d over (B, n) per channel: the same numbers as the reference's row form
        (rows = points, BatchNorm1d over the rows) without flatten / permute copies and without handing
        (B*n) x 35 -> 32 -> ... -> 1 products to a GEMM library (hipBLASLt spends ~4 ms per step on them at
        B*n = 192000).  -> (B*n, 1)"""
        import os
        from amcontrast3d_amd.ops import BatchNormAct, BatchNormSigmoid, mixed_precision, pointwise_conv
        from openpoints.models.layers.blocks import _fusable_bn
        x = torch.cat((p.transpose(1, 2), f), dim=1).contiguous()
        mods = list(tower)
        fuse = not os.environ.get("AMC3D_NO_BN_SIGMOID")
        skip = False
        for i, mod in enumerate(mods):
            if skip:  # the Sigmoid that the BatchNorm in front of it has already applied
                skip = False
                continue
            if isinstance(mod, nn.Linear):
                x = pointwise_conv(x, mod.weight.unsqueeze(-1), mod.bias, mixed_precision())
            elif isinstance(mod, nn.BatchNorm1d):
                if _fusable_bn(mod, x) and fuse and i + 1 < len(mods) and type(mods[i + 1]) is nn.Sigmoid:
                    x = BatchNormSigmoid.apply(x, mod.weight, mod.bias, mod.eps, mod)[0]  # BatchNorm + Sigmoid, two launches
                    skip = True
                else:
                    x = BatchNormAct.apply(x, mod.weight, mod.bias, mod.eps, False, mod)[0] if _fusable_bn(mod, x) else mod(x)
            else:  # Dropout (p = 0 in the shipped configs), Sigmoid
                x = mod(x)
        return x.reshape(-1, 1)  # (B,1,n) -> (B*n,1): same memory order as the flattened rows

    def forward(self, p, f):
        """p (B,n,3) with f (B,D,n), or already flattened p (m,3) with f (m,D) -> a (m,1) [, a_map (m,D)]"""
        if (p.dim() == 3 and f.dim() == 3 and f.is_cuda and f.dtype == torch.float32 and not self.map
                and f.shape[1] in self.dim):
            return self._tower_channel_major(getattr(self, f'layer_{self.dim.index(f.shape[1])}'), p, f)
        if not (p.dim() == 2 and f.dim() == 2):
            p = torch.flatten(p, start_dim=0, end_dim=1)
            f = torch.flatten(f.permute(0, 2, 1), start_dim=0, end_dim=1)
        x = torch.cat((p, f), dim=1)
        for s, d in enumerate(self.dim):  # the tower is chosen by the feature width, first match (concatenation.py:176-195)
            if f.shape[1] == d:
                tower = getattr(self, f'layer_{s}')
                if self.map:
                    # the reference evaluates the tower twice here (two BatchNorm running-stat updates per call)
                    return tower(x), getattr(self, f'map_{s}')(tower(x))
                return tower(x)
        return None
