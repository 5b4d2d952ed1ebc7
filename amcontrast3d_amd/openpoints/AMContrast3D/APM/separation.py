"""Position-only ablation variants of the Ambiguity Prediction Module: a = APM(p).

Drop-in for openpoints/AMContrast3D/APM/separation.py:
    APM_p        :12-60    a per-point MLP regressor on the coordinates alone
    KNN          :63-71    k nearest neighbours of every point among ALL points handed in, self excluded
    APM_p_Group  :74-129   1x1 convolutions over [p_i ; |p_i - p_j| for the k-1 nearest j], a 3 -> 1 regressor, softmax
    APM_p_Graph  :167-242  a graph convolution over the same neighbourhoods

What a switch-over user should know about the reference's own state of these three (they are listed as alternatives in
``cfgs/*/AMContrast3D-MM.yaml:40``; the shipped configs use APM_pf_ConCate):
  * ``APM_p.__init__`` reads ``self.drop_rate`` without ever setting it (separation.py:29), so the reference class cannot be
    constructed (AttributeError).  Built here as evidently meant: the rates are the ``dropout`` argument.  Same Sequential
    positions, hence the state-dict keys ``layers.{0,4,8,12,16,20}`` Linear / ``layers.{2,6,10,14,18,21}`` BatchNorm1d.
  * ``APM_p_Group`` flattens the batch to one point list and searches it with a single offset (separation.py:64-65), so a
    point's neighbours may lie in another cloud of the batch; and its ``F.softmax(out)`` has no ``dim``: on the (B, n, 1)
    tensor torch's legacy rule picks dim 0, the softmax runs ACROSS THE CLOUDS of the batch.  Both kept.
    Keys: ``conv.{0,3,6}`` Conv1d (no bias), ``conv.{1,4,7}`` BatchNorm1d, ``regressor``.
  * ``APM_p_Graph.__init__`` names ``GCNConv`` (torch_geometric), which separation.py never imports: NameError at
    construction.  torch_geometric is not part of this image either; the class raises NotImplementedError saying so.
The search is ``pointops.knnquery`` (csrc/knn.hip through ops.KNNQuery): GPU tensors only, like the reference's.
"""
from typing import List

import torch
import torch.nn as nn

from openpoints.models.build import MODELS
from openpoints.cpp.pointops.functions import pointops


@MODELS.register_module()
class APM_p(nn.Module):
    def __init__(self, feature_dim: List[int] = [64, 128, 256, 512], linear_mapping: bool = True,
                 cross_attention: bool = False, feat_concate: bool = True, channel: List[int] = [64, 32, 16, 8, 4, 2],
                 dropout: List[float] = [0.2, 0, 0, 0, 0, 0], nsample_k: int = 12, threshold: float = 0.7,
                 threshold_max: float = 1.0, gamma: float = 0.5, fusion: str = 'MIN', att_dim: int = 3):
        super().__init__()
        self.drop_rate = list(dropout)
        mods, prev = [], 3
        for width, rate in zip(channel[:5], self.drop_rate[:5]):  # five hidden layers; channel[5] is unused there too
            mods += [nn.Linear(prev, width), nn.Dropout(rate), nn.BatchNorm1d(width), nn.Sigmoid()]
            prev = width
        mods += [nn.Linear(prev, 1), nn.BatchNorm1d(1), nn.Sigmoid()]
        self.layers = nn.Sequential(*mods)

    def forward(self, p):
        return self.layers(torch.flatten(p, start_dim=0, end_dim=1))


def KNN(p, k):
    """p (m,3) -> indices (m,k-1) and positions (m,k-1,3) of the k-1 nearest OTHER points (the nearest of the k found is
    the point itself and is dropped), all m points taken as one cloud"""
    o = torch.tensor([p.shape[0]], dtype=torch.int32, device=p.device)
    idx, _ = pointops.knnquery(k, p, p, o, o)
    idx = idx[..., 1:].contiguous()
    return idx, p[idx.reshape(-1).long(), :].view(idx.shape[0], k - 1, p.shape[1])


def _relative_positions(p, k):
    """(m,3) -> (m,k,3): row 0 the point, rows 1..k-1 |p_i - p_j| over its neighbours"""
    _, nbr = KNN(p, k)
    centre = p.unsqueeze(1)
    return torch.cat([centre, (centre - nbr).abs()], dim=1)


@MODELS.register_module()
class APM_p_Group(nn.Module):
    def __init__(self, feature_dim: List[int] = [64, 128, 256, 512], linear_mapping: bool = True,
                 cross_attention: bool = False, feat_concate: bool = True, channel: List[int] = [64, 32, 16, 8, 4, 2],
                 dropout: List[float] = [0.2, 0, 0, 0, 0, 0], nsample_k: int = 12, threshold: float = 0.7,
                 threshold_max: float = 1.0, gamma: float = 0.5, fusion: str = 'MIN', att_dim: int = 3):
        super().__init__()
        self.k = nsample_k
        self.in_channels = nsample_k * 3
        mods, prev = [], self.in_channels
        for width in (18, 9, 3):
            mods += [nn.Conv1d(prev, width, kernel_size=1, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            prev = width
        self.conv = nn.Sequential(*mods)
        self.regressor = nn.Linear(3, 1)

    def forward(self, p):
        B = p.shape[0]
        rel = _relative_positions(torch.flatten(p, start_dim=0, end_dim=1), self.k)   # (m, k, 3)
        h = self.conv(rel.reshape(B, -1, self.k * 3).transpose(1, 2))                  # (B, 3, n)
        out = self.regressor(h.transpose(1, 2))                                        # (B, n, 1)
        out = torch.softmax(out, dim=0)  # what F.softmax(out) without dim does on a 3-d tensor (separation.py:126)
        return torch.flatten(out, start_dim=0, end_dim=1)


@MODELS.register_module()
class APM_p_Graph(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("APM_p_Graph needs torch_geometric's GCNConv, which the reference's separation.py uses "
                                  "without importing (its own class fails with NameError) and which this image does not have; "
                                  "use APM_pf_ConCate (the shipped choice) or one of APM_p / APM_p_Group / APM_pf_CrossAtt / "
                                  "APM_pp_SelfAtt")
