"""Neighbourhood grouping layers on the gfx950 kernels.

API mirror of models/layers/group.py: ``ball_query``, ``grouping_operation``,
``gather_operation``, ``QueryAndGroup`` (:206-255), ``GroupAll`` (:258-272),
``get_aggregation_feautres`` (:323-335, name kept as spelled there),
``create_grouper`` (:338-352), ``torch_grouping_operation`` (:120-137).
"""
import copy
import logging

import torch
import torch.nn as nn

from amcontrast3d_amd.ops import ball_query, gather_operation, grouping_operation


def torch_grouping_operation(features, idx):
    """pure-torch gather: (B,C,N), (B,npoint,nsample) -> (B,C,npoint,nsample)"""
    B, C = features.shape[:2]
    flat = idx.reshape(B, 1, -1).expand(-1, C, -1).long()
    return features.gather(2, flat).reshape(B, C, idx.shape[1], idx.shape[2])


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, relative_xyz=True, normalize_dp=False, normalize_by_std=False,
                 normalize_by_allstd=False, normalize_by_allstd2=False, return_only_idx=False, **kwargs):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.normalize_dp = normalize_dp
        self.normalize_by_std = normalize_by_std
        self.normalize_by_allstd = normalize_by_allstd
        self.normalize_by_allstd2 = normalize_by_allstd2
        assert self.normalize_dp + self.normalize_by_std + self.normalize_by_allstd < 2
        self.relative_xyz = relative_xyz
        self.return_only_idx = return_only_idx

    def query(self, query_xyz, support_xyz):
        """neighbour indices (B,npoint,nsample) int32 -- geometry only"""
        return ball_query(self.radius, self.nsample, support_xyz, query_xyz)

    def relative_positions(self, idx, query_xyz, support_xyz):
        """(B,3,npoint,nsample) neighbour offsets, divided by the radius if normalize_dp -- geometry only"""
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz = grouped_xyz - query_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_dp:
                grouped_xyz /= self.radius
        return grouped_xyz

    def forward(self, query_xyz, support_xyz, features=None, geom=None):
        """query (B,npoint,3), support (B,N,3), features (B,C,N)
        -> relative positions (B,3,npoint,nsample), grouped features (B,C,npoint,nsample).
        `geom` = {'idx', 'dp'} precomputed by query()/relative_positions() (they do not depend on the
        features, so a caller may prepare them ahead of time / on another stream)."""
        idx = geom['idx'] if geom is not None else self.query(query_xyz, support_xyz)
        if self.return_only_idx:
            return idx
        grouped_xyz = geom['dp'] if geom is not None else self.relative_positions(idx, query_xyz, support_xyz)
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features


class GroupAll(nn.Module):
    def forward(self, new_xyz, xyz, features=None, geom=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        grouped_features = features.unsqueeze(2) if features is not None else None
        return grouped_xyz, grouped_features


def get_aggregation_feautres(p, dp, f, fj, feature_type='dp_fj'):
    if feature_type == 'dp_fj':
        return torch.cat([dp, fj], 1)
    if feature_type == 'dp_fj_df':
        return torch.cat([dp, fj, fj - f.unsqueeze(-1)], 1)
    if feature_type == 'pi_dp_fj_df':
        df = fj - f.unsqueeze(-1)
        pi = p.transpose(1, 2).unsqueeze(-1).expand(-1, -1, -1, df.shape[-1])
        return torch.cat([pi, dp, fj, df], 1)
    if feature_type == 'dp_df':
        return torch.cat([dp, fj - f.unsqueeze(-1)], 1)
    return fj


def create_grouper(group_args):
    args = copy.deepcopy(group_args)
    method = args.pop('NAME', 'ballquery')
    radius = args.pop('radius', 0.1)
    nsample = args.pop('nsample', 20)
    logging.info(group_args)
    if nsample is None:
        return GroupAll()
    if method == 'ballquery':
        return QueryAndGroup(radius, nsample, **args)
    raise NotImplementedError(f"grouper {method!r}: only 'ballquery' is on the AMContrast3D path")
