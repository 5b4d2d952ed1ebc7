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

    def forward(self, p, f):
        """p (B,n,3) with f (B,D,n), or already flattened p (m,3) with f (m,D) -> a (m,1) [, a_map (m,D)]"""
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
