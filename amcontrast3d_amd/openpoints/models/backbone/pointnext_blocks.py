"""Building blocks of the PointNeXt backbone on the gfx950 kernels: the geometry / feature split.

Classes the reference defines in openpoints/models/backbone/pointnext_AA.py and that keep their names, constructor
keywords, attribute names and module nesting (hence state-dict keys) here:

    LocalAggregation                 pointnext_AA.py:22-73
    SetAbstraction                   :76-170
    FeaturePropogation               :173-226   (spelling kept: it is in checkpoints' class paths)
    InvResMLP / ResBlock             :229-308

Every block has a coordinate-only ``plan*`` method (FPS picks, ball-query indices, relative positions, 3-NN indices
and weights) and a ``forward`` that consumes such a plan, building it itself when none is handed in -- one code
path whether or not a trainer prepares the geometry of the next batch ahead of time.  They are re-exported by
pointnext_AA.py, where the reference's importers expect them.
"""
import logging
from typing import List

import torch
import torch.nn as nn

from ..layers import (CHANNEL_MAP, create_act, create_convblock1d, create_convblock2d, create_grouper,
                      feature_propagation_first_block, gather_operation,
                      furthest_point_sample, fused_first_block, fused_first_conv, fused_local_aggregation,
                      get_aggregation_feautres,
                      random_sample, run_convblocks,
                      three_interpolate, three_nn)


def _moments(convs, feature_type, idx, dp, n_support, csr=None):
    """geometry moments of a neighbourhood layer (the convolve-before-gather kernels need them), else None"""
    if feature_type != 'dp_fj' or not idx.is_cuda or len(convs) < 1:
        return None
    from amcontrast3d_amd import ops
    if csr is not None:
        return ops.group_moments_csr(idx, dp, n_support, (csr['start'], csr['edge']))
    return ops.group_moments(idx, dp, n_support)


def _csr(convs, feature_type, idx, n_support, dp=None):
    """reverse adjacency of the query for layers whose backward sums dense per-position gradients into the source points
    (the multi-layer SetAbstraction MLP of PointNeXt-S): gathers over sorted edge lists instead of float atomics"""
    import os
    if feature_type != 'dp_fj' or not idx.is_cuda or len(convs) < 2 or os.environ.get("AMC3D_NO_CSR"):
        return None
    from amcontrast3d_amd import ops
    start, edge = ops.group_csr(idx, n_support)
    csr = {'start': start, 'edge': edge}
    if dp is not None and dp.dtype == torch.float32 and not os.environ.get("AMC3D_NO_CSR_DP"):
        csr['edge_dp'] = ops.group_csr_dp(idx, dp, edge)  # (dp, position) per edge in list order: one stream for the backward's gather
    return csr


def get_reduction_fn(reduction):
    reduction = 'mean' if reduction.lower() == 'avg' else reduction
    assert reduction in ['sum', 'max', 'mean']
    if reduction == 'max':
        return lambda x: torch.max(x, dim=-1, keepdim=False)[0]
    if reduction == 'mean':
        return lambda x: torch.mean(x, dim=-1, keepdim=False)
    return lambda x: torch.sum(x, dim=-1, keepdim=False)


class LocalAggregation(nn.Module):
    """Grouped MLP over the neighbourhood of every point of one set, then pooling."""

    def __init__(self, channels: List[int], norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'},
                 group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16}, conv_args=None,
                 feature_type='dp_fj', reduction='max', last_act=True, **kwargs):
        super().__init__()
        if kwargs:
            logging.warning(f"kwargs: {kwargs} are not used in {__class__.__name__}")
        channels[0] = CHANNEL_MAP[feature_type](channels[0])
        last = len(channels) - 2
        self.convs = nn.Sequential(*[
            create_convblock2d(channels[i], channels[i + 1], norm_args=norm_args,
                               act_args=None if (i == last and not last_act) else act_args, **(conv_args or {}))
            for i in range(len(channels) - 1)])
        self.grouper = create_grouper(group_args)
        self.reduction = reduction.lower()
        self.pool = get_reduction_fn(self.reduction)
        self.feature_type = feature_type

    @torch.no_grad()
    def plan(self, p):
        """coordinate-only part: neighbour indices and relative positions of the self query"""
        if not hasattr(self.grouper, 'query'):
            return None
        idx = self.grouper.query(p, p)
        dp = self.grouper.relative_positions(idx, p, p)
        return {'idx': idx, 'dp': dp, 'mom': _moments(self.convs, self.feature_type, idx, dp, p.shape[1])}

    def forward(self, pf, geom=None):
        p, f = pf
        if geom is None:
            geom = self.plan(p)
        if self.reduction == 'max':
            y = fused_local_aggregation(self.convs, f, geom, self.feature_type)
            if y is not None:  # conv on the source points, then gather + BN + ReLU + max in one pass
                return y
        pre = fused_first_conv(self.convs, f, geom, self.feature_type)
        if pre is not None:  # gather + concat + first conv in one MFMA kernel
            y = run_convblocks(self.convs, None, pool_max=self.reduction == 'max', pre=pre)
            return y if self.reduction == 'max' else self.pool(y)
        dp, fj = self.grouper(p, p, f, geom=geom)
        fj = get_aggregation_feautres(p, dp, f, fj, self.feature_type)
        if self.reduction == 'max':
            return run_convblocks(self.convs, fj, pool_max=True)
        return self.pool(run_convblocks(self.convs, fj))


class SetAbstraction(nn.Module):
    """FPS -> ball-query grouping -> grouped MLP -> max over the neighbourhood
    (-> + skip connection, ReLU).  With ``is_head`` it is the point-wise stem MLP."""

    def __init__(self, in_channels, out_channels, layers=1, stride=1,
                 group_args={'NAME': 'ballquery', 'radius': 0.1, 'nsample': 16},
                 norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'}, conv_args=None, sampler='fps',
                 feature_type='dp_fj', use_res=False, is_head=False, **kwargs):
        super().__init__()
        self.stride = stride
        self.is_head = is_head
        self.all_aggr = not is_head and stride == 1
        self.use_res = use_res and not self.all_aggr and not self.is_head
        self.feature_type = feature_type

        mid = out_channels // 2 if stride > 1 else out_channels
        channels = [in_channels] + [mid] * (layers - 1) + [out_channels]
        if not is_head:
            channels[0] = CHANNEL_MAP[feature_type](channels[0])

        if self.use_res:
            self.skipconv = create_convblock1d(in_channels, channels[-1], norm_args=None, act_args=None) \
                if in_channels != channels[-1] else nn.Identity()
            self.act = create_act(act_args)

        make = create_convblock1d if is_head else create_convblock2d
        last = len(channels) - 2
        self.convs = nn.Sequential(*[
            make(channels[i], channels[i + 1], norm_args=None if is_head else norm_args,
                 act_args=None if (i == last and (self.use_res or is_head)) else act_args, **(conv_args or {}))
            for i in range(len(channels) - 1)])

        if not is_head:
            if self.all_aggr:
                group_args.nsample = None
                group_args.radius = None
            self.grouper = create_grouper(group_args)
            self.pool = lambda x: torch.max(x, dim=-1, keepdim=False)[0]
            if sampler.lower() == 'fps':
                self.sample_fn = furthest_point_sample
            elif sampler.lower() == 'random':
                self.sample_fn = random_sample

    @torch.no_grad()
    def plan_sample(self, p):
        """FPS picks and the sub-sampled cloud (the serial part of the geometry)"""
        if self.is_head or self.all_aggr:
            return {'fps_idx': None, 'new_p': p}
        idx32 = self.sample_fn(p, p.shape[1] // self.stride)
        idx = idx32.long()
        g = {'fps_idx': idx, 'new_p': torch.gather(p, 1, idx.unsqueeze(-1).expand(-1, -1, 3))}
        if idx32.dtype == torch.int32:
            g['fps_idx32'] = idx32.contiguous()
        return g

    @torch.no_grad()
    def plan_group(self, p, g):
        """neighbour indices and relative positions around the sampled points (in place into g)"""
        if not self.is_head and hasattr(self.grouper, 'query'):
            g['idx'] = self.grouper.query(g['new_p'], p)
            g['dp'] = self.grouper.relative_positions(g['idx'], g['new_p'], p)
            g['csr'] = _csr(self.convs, self.feature_type, g['idx'], p.shape[1], g['dp'])
            g['mom'] = _moments(self.convs, self.feature_type, g['idx'], g['dp'], p.shape[1], g['csr'])
            if getattr(self, 'use_res', False) and g.get('fps_idx32') is not None and g['fps_idx32'].is_cuda:
                # do the picks of some cloud repeat an index?  (torch.gather's backward sums over repeats; the fused residual's
                # backward scatters with plain stores when they do not: csrc/sa_res.hip)
                from amcontrast3d_amd import ops
                g['fps_dup'] = ops.index_duplicates(g['fps_idx32'], p.shape[1])
        return g

    def plan(self, p):
        """coordinate-only part: FPS picks, the sub-sampled cloud, neighbour indices, relative positions"""
        return self.plan_group(p, self.plan_sample(p))

    def forward(self, pf, geom=None):
        p, f = pf
        if self.is_head:
            return p, run_convblocks(self.convs, f)
        if geom is None:
            geom = self.plan(p)
        idx, new_p = geom['fps_idx'], geom['new_p']
        fi = None
        res_fused = self._fused_residual(f, geom)
        if not res_fused and (self.use_res or 'df' in self.feature_type):
            import os
            if (f.is_cuda and f.dtype == torch.float32 and geom.get('fps_idx32') is not None
                    and os.environ.get("AMC3D_OWN_GATHER")):
                # gather_points kernels (the reference's GatherOperation): measured 0.24 ms/step slower than torch.gather
                # here (its backward scatters with float atomics, torch's index-add over sorted FPS picks does not)
                fi = gather_operation(f.contiguous(), geom['fps_idx32'])
            else:
                fi = torch.gather(f, -1, idx.unsqueeze(1).expand(-1, f.shape[1], -1))
            if self.use_res:
                identity = run_convblocks((self.skipconv,), fi)
        fused = fused_local_aggregation(self.convs, f, geom, self.feature_type)
        x1 = fused_first_block(self.convs, f, geom, self.feature_type) if fused is None else None
        pre = fused_first_conv(self.convs, f, geom, self.feature_type) if (fused is None and x1 is None) else None
        if fused is not None:  # single conv layer: conv on the source points, then gather + BN + ReLU + max in one pass
            f = fused
        elif x1 is not None:   # two-layer MLP: first conv + BN + ReLU on the source points / one gather pass, then the rest
            f = run_convblocks(list(self.convs)[1:], x1, pool_max=True, activated=True)
        elif pre is not None:  # gather + concat + first conv in one MFMA kernel
            f = run_convblocks(self.convs, None, pool_max=True, pre=pre)
        else:
            dp, fj = self.grouper(new_p, p, f, geom=geom if 'idx' in geom else None)
            fj = get_aggregation_feautres(new_p, dp, fi, fj, feature_type=self.feature_type)
            f = run_convblocks(self.convs, fj, pool_max=True)  # conv/BN/ReLU stack + max over the neighbours
        if res_fused:  # gather at the FPS picks + skip conv + bias + add + ReLU as one kernel (csrc/sa_res.hip)
            from amcontrast3d_amd import ops
            conv = self.skipconv[0]
            f = ops.sa_residual(f, pf[1], geom['fps_idx32'], conv.weight, conv.bias, geom.get('fps_dup'))
        elif self.use_res:
            f = self.act(f + identity)
        return new_p, f

    def _fused_residual(self, f, geom):
        """True when the residual branch (pointnext_AA.py:157-168) has the form the fused kernels cover: a bare 1x1 Conv1d
        as skip conv, ReLU, fp32 features on the GPU and int32 FPS picks in the plan"""
        import os
        if (not self.use_res or 'df' in self.feature_type or os.environ.get("AMC3D_NO_SA_RESIDUAL")
                or geom.get('fps_idx32') is None or not f.is_cuda or f.dtype != torch.float32 or f.dim() != 3):
            return False
        sk = self.skipconv
        if not isinstance(sk, nn.Sequential) or len(sk) != 1 or not isinstance(sk[0], nn.Conv1d) or type(self.act) is not nn.ReLU:
            return False
        conv = sk[0]
        return (conv.kernel_size == (1,) and conv.stride == (1,) and conv.groups == 1 and conv.padding == (0,)
                and conv.in_channels == f.shape[1] and conv.weight.dtype == torch.float32)


class FeaturePropogation(nn.Module):
    """PointNet++ feature propagation: 3-NN inverse-distance interpolation of the
    coarse features onto the fine set, concat with the skip features, point-wise MLP."""

    def __init__(self, mlp, upsample=True, norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'}):
        super().__init__()
        if not upsample:
            self.linear2 = nn.Sequential(nn.Linear(mlp[0], mlp[1]), nn.ReLU(inplace=True))
            mlp[1] *= 2
            self.linear1 = nn.Sequential(*[
                create_convblock1d(mlp[i], mlp[i + 1], norm_args=norm_args, act_args=act_args)
                for i in range(1, len(mlp) - 1)])
        else:
            self.convs = nn.Sequential(*[
                create_convblock1d(mlp[i], mlp[i + 1], norm_args=norm_args, act_args=act_args)
                for i in range(len(mlp) - 1)])
        self.pool = lambda x: torch.mean(x, dim=-1, keepdim=False)

    @staticmethod
    @torch.no_grad()
    def plan(p1, p2):
        """coordinate-only part: the 3 nearest coarse points of every fine point and their
        inverse-distance weights (upsampling.py:97-100)"""
        dist, idx = three_nn(p1, p2)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        return {'idx': idx, 'weight': weight}

    def forward(self, pf1, pf2=None, geom=None):
        if pf2 is None:  # global branch (not used by the segmentation decoder)
            _, f = pf1
            g = self.linear2(self.pool(f))
            return self.linear1(torch.cat((f, g.unsqueeze(-1).expand(-1, -1, f.shape[-1])), dim=1))
        p1, f1 = pf1
        p2, f2 = pf2
        if geom is None:
            geom = self.plan(p1, p2)
        first = feature_propagation_first_block(self.convs[0], f1, f2, geom) if len(self.convs) else None
        if first is not None:  # the first conv on both branches before the interpolation: no concatenated tensor
            return run_convblocks(list(self.convs)[1:], first)
        up = three_interpolate(f2, geom['idx'], geom['weight'])
        return run_convblocks(self.convs, up if f1 is None else torch.cat((f1, up), dim=1))


class InvResMLP(nn.Module):
    """LocalAggregation (C->C) + point-wise C->expansion*C->C with a residual."""

    def __init__(self, in_channels, norm_args=None, act_args=None,
                 aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, group_args={'NAME': 'ballquery'},
                 conv_args=None, expansion=1, use_res=True, num_posconvs=2, less_act=False, **kwargs):
        super().__init__()
        self.use_res = use_res
        mid_channels = int(in_channels * expansion)
        self.convs = LocalAggregation([in_channels, in_channels], norm_args=norm_args,
                                      act_args=act_args if num_posconvs > 0 else None, group_args=group_args,
                                      conv_args=conv_args, **aggr_args, **kwargs)
        if num_posconvs < 1:
            channels = []
        elif num_posconvs == 1:
            channels = [in_channels, in_channels]
        else:
            channels = [in_channels, mid_channels, in_channels]
        last = len(channels) - 2
        self.pwconv = nn.Sequential(*[
            create_convblock1d(channels[i], channels[i + 1], norm_args=norm_args,
                               act_args=act_args if (i != last and not less_act) else None, **(conv_args or {}))
            for i in range(len(channels) - 1)])
        self.act = create_act(act_args)

    def plan(self, p):
        return self.convs.plan(p)

    def forward(self, pf, geom=None):
        p, f = pf
        identity = f
        agg = self.convs([p, f], geom=geom)
        if (self.use_res and type(self.act) is nn.ReLU and len(self.pwconv) and agg.shape[-1] == identity.shape[-1]
                and agg.shape[1] == identity.shape[1]):
            # `f += identity; act(f)` inside the last BatchNorm's kernels where the block has that form
            return [p, run_convblocks(self.pwconv, agg, residual=identity)]
        f = run_convblocks(self.pwconv, agg)
        if f.shape[-1] == identity.shape[-1] and self.use_res:
            f += identity
        return [p, self.act(f)]


class ResBlock(nn.Module):
    def __init__(self, in_channels, norm_args=None, act_args=None,
                 aggr_args={'feature_type': 'dp_fj', "reduction": 'max'}, group_args={'NAME': 'ballquery'},
                 conv_args=None, expansion=1, use_res=True, **kwargs):
        super().__init__()
        self.use_res = use_res
        mid_channels = in_channels * expansion
        self.convs = LocalAggregation([in_channels, in_channels, mid_channels, in_channels], norm_args=norm_args,
                                      act_args=None, group_args=group_args, conv_args=conv_args, **aggr_args,
                                      **kwargs)
        self.act = create_act(act_args)

    def plan(self, p):
        return self.convs.plan(p)

    def forward(self, pf, geom=None):
        p, f = pf
        identity = f
        f = self.convs([p, f], geom=geom)
        if f.shape[-1] == identity.shape[-1] and self.use_res:
            f += identity
        return [p, self.act(f)]


_BLOCKS = {'InvResMLP': InvResMLP, 'ResBlock': ResBlock}
