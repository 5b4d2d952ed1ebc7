"""Attention-flavoured ablation variants of the Ambiguity Prediction Module.

Drop-in for openpoints/AMContrast3D/APM/attention.py:
    Attention        :10-35    single-head dot-product attention with bias-free query / key / value maps
    APM_pf_CrossAtt  :38-123   a = tower_s(Attn(ext_s(p), f)): the query comes from the positions, keys / values from the features
    APM_pp_SelfAtt   :126-161  a = layers(Attn(p, p))

Behaviour that is kept because a switch-over user's numbers depend on it (none of it is an optimisation target: these are
ablation rows of the reference's config comment, ``cfgs/*/AMContrast3D-MM.yaml:40``, not its shipped choice):
  * the attention layer is NOT a sub-module: the reference builds a fresh, randomly initialised one inside every forward
    (attention.py:107, 157), so it is never trained, never in the state dict, and every call draws three weight
    initialisations from torch's global generator, in the order query, key, value;
  * the inputs arrive flattened to rows (m, D) and ``Attention.forward`` takes ``x.shape[0]`` for the batch, so every point is
    its own one-token sequence: scores are (m, 1, 1), the softmax over one key is 1 and the output equals value(y), shaped
    (m, 1, dv).  The general formula is evaluated anyway (same operations, same results for any other input shape);
  * the towers end in BatchNorm1d(1) on (m, 1, 1) and return (m, 1, 1); no mapped embedding is returned, so
    ``linear_mapping: True`` fails in BaseSeg_M_AMContrast3D at the tuple unpacking, as it does in the reference.
State-dict keys: ``layer_{s}.{0,2,4}`` Linear, ``layer_{s}.5`` BatchNorm1d, ``ext_{s}.0`` Linear (CrossAtt);
``layers.{0,2,4}`` Linear, ``layers.5`` BatchNorm1d (SelfAtt).
The reference moves the fresh layer ``.to('cuda')``; here it follows its input's device (the CPU tests run it).
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from openpoints.models.build import MODELS


class Attention(nn.Module):
    def __init__(self, in_dim: int = 64, dk: int = 3, dv: int = 64):
        super().__init__()
        self.in_dim = in_dim
        self.query = nn.Linear(in_dim, dk, bias=False)
        self.key = nn.Linear(in_dim, dk, bias=False)
        self.value = nn.Linear(in_dim, dv, bias=False)

    def forward(self, x, y):
        n = x.shape[0]
        x, y = x.view(n, -1, self.in_dim), y.view(n, -1, self.in_dim)
        scores = torch.matmul(self.query(x), self.key(y).transpose(-2, -1)) / (self.in_dim ** 0.5)
        return torch.bmm(F.softmax(scores, dim=-1), self.value(y))


def _sigmoid_tower(cin: int, channel: List[int]) -> nn.Sequential:
    """Linear, Sigmoid, Linear, Sigmoid, Linear -> 1, BatchNorm1d(1), Sigmoid  (attention.py:58-66)"""
    return nn.Sequential(nn.Linear(cin, channel[0]), nn.Sigmoid(), nn.Linear(channel[0], channel[1]), nn.Sigmoid(),
                         nn.Linear(channel[1], 1), nn.BatchNorm1d(1), nn.Sigmoid())


def _rows(p, f=None):
    p = torch.flatten(p, start_dim=0, end_dim=1)
    if f is None:
        return p
    return p, torch.flatten(f.permute(0, 2, 1), start_dim=0, end_dim=1)


@MODELS.register_module()
class APM_pf_CrossAtt(nn.Module):
    def __init__(self, feature_dim: List[int] = [64, 128, 256, 512], linear_mapping: bool = True,
                 cross_attention: bool = False, feat_concate: bool = True, channel: List[int] = [32, 16, 8, 4, 2],
                 dropout: List[float] = [0, 0, 0, 0, 0], nsample_k: int = 12, threshold: float = 0.7,
                 threshold_max: float = 1.0, gamma: float = 0.5, fusion: str = 'MIN', att_dim: int = 3):
        super().__init__()
        assert len(feature_dim) == 4
        self.dim = list(feature_dim)
        self.map = linear_mapping
        self.drop_rate = list(dropout)
        self.mask_dim = att_dim
        for s, d in enumerate(self.dim):  # registration order of the reference: the four towers, then the four lifts
            setattr(self, f'layer_{s}', _sigmoid_tower(d, channel))
        for s, d in enumerate(self.dim):
            setattr(self, f'ext_{s}', nn.Sequential(nn.Linear(3, d), nn.Sigmoid()))

    def forward(self, p, f):
        p, f = _rows(p, f)
        width = f.shape[1]
        cross_layer = Attention(width, self.mask_dim, width).to(f.device)  # fresh weights every call (see the module docstring)
        for s, d in enumerate(self.dim):  # first tower whose width matches
            if width == d:
                return getattr(self, f'layer_{s}')(cross_layer(getattr(self, f'ext_{s}')(p), f))
        return None


@MODELS.register_module()
class APM_pp_SelfAtt(nn.Module):
    def __init__(self, feature_dim: List[int] = [64, 128, 256, 512], linear_mapping: bool = True,
                 cross_attention: bool = False, feat_concate: bool = True, channel: List[int] = [32, 16, 8, 4, 2],
                 dropout: List[float] = [0, 0, 0, 0, 0], nsample_k: int = 12, threshold: float = 0.7,
                 threshold_max: float = 1.0, gamma: float = 0.5, fusion: str = 'MIN', att_dim: int = 3):
        super().__init__()
        self.dim = list(feature_dim)
        self.map = linear_mapping
        self.drop_rate = list(dropout)
        self.mask_dim = att_dim
        self.layers = _sigmoid_tower(3, channel)

    def forward(self, p):
        p = _rows(p)
        cross_layer = Attention(p.shape[1], self.mask_dim, p.shape[1]).to(p.device)
        return self.layers(cross_layer(p, p))
