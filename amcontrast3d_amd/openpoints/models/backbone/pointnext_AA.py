"""PointNeXt encoder/decoder with the AMContrast3D stage bookkeeping.

Drop-in for openpoints/models/backbone/pointnext_AA.py: same registered class
names, constructor keywords, attribute names, module nesting (hence state-dict
keys) and return structures:

    PointNextEncoder_AMContrast3D    pointnext_AA.py:311-471
    PointNextDecoder_AMContrast3D    :475-527
    LocalAggregation, SetAbstraction, FeaturePropogation, InvResMLP, ResBlock   re-exported from pointnext_blocks.py

All neighbour search, sampling, grouping and interpolation runs on the gfx950
kernels in amcontrast3d_amd (no torch fallback).

Geometry / feature split.  Everything that depends only on the coordinates -- FPS picks,
ball-query indices, relative positions, 3-NN indices and weights -- is computed by the
``plan*`` methods and consumed by ``forward``; ``forward`` builds the plan itself when the
batch dict carries none (key ``'_geometry'``), so there is one code path.  A trainer can build
the plan of the NEXT batch on a side stream while the current batch trains (bench.py does):
FPS is a serial chain that occupies 8 of the 256 CUs.
"""
import logging
from typing import List

import torch
import torch.nn as nn

from amcontrast3d_amd.ops import point_major_rows

from ..build import MODELS
from .pointnext_blocks import (_BLOCKS, FeaturePropogation, InvResMLP, LocalAggregation, ResBlock,  # noqa: F401
                               SetAbstraction, get_reduction_fn)


_OFFSETS = {}


def _set_embedding(stage, f):
    """stage['f_out'] = rows of the channel-major embedding f (B, C, n); deferred where the stage entry can defer it"""
    if isinstance(stage, _Stage):
        stage.set_features(f)
    else:
        stage['f_out'] = point_major_rows(f)


class _Stage(dict):
    """One entry of stageACE_list['down'] / ['up']: {'p_out', 'f_out', 'offset'} (pointnext_AA.py:458-462).  The
    encoder's 'f_out' -- a transposed copy of its features -- is overwritten by the decoder before anything reads it
    ('up' IS 'down', :464-465, 518-519), so it is made only if somebody asks for it first."""

    def __init__(self, p_out, features, offset):
        super().__init__(p_out=p_out, offset=offset)
        self._features = features

    def _materialise(self):
        if self._features is not None and not dict.__contains__(self, 'f_out'):
            from amcontrast3d_amd.ops import point_major_rows
            dict.__setitem__(self, 'f_out', point_major_rows(self._features))
        self._features = None

    def set_features(self, features):
        """the decoder's embedding of this resolution, channel-major (B, C, n): 'f_out' -- its (B*n, C) rows,
        pointnext_AA.py:518-519 -- is made when somebody reads it; the fused loss stage takes channel_major() instead"""
        dict.pop(self, 'f_out', None)
        self._features = features

    def channel_major(self):
        return self._features

    def __getitem__(self, key):
        if key == 'f_out':
            self._materialise()
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        if key == 'f_out':
            self._features = None
        dict.__setitem__(self, key, value)

    def __contains__(self, key):
        return key == 'f_out' or dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def __len__(self):
        return 3


def _segment_offset(n, device):
    """int32 [n] on `device` -- the single-segment offset of a flattened batch.  Cached per
    (n, device): the reference builds it with IntTensor([n]).cuda() every stage and step
    (pointnext_AA.py:461), a pageable host-to-device copy that also cannot be graph-captured."""
    key = (int(n), str(device))
    t = _OFFSETS.get(key)
    if t is None:
        t = _OFFSETS[key] = torch.tensor([int(n)], dtype=torch.int32, device=device)
    return t


@MODELS.register_module()
class PointNextEncoder_AMContrast3D(nn.Module):
    def __init__(self, in_channels: int = 4, width: int = 32, blocks: List[int] = [1, 4, 7, 4, 4],
                 strides: List[int] = [4, 4, 4, 4], block='InvResMLP', nsample=32, radius=0.1,
                 aggr_args: dict = {'feature_type': 'dp_fj', "reduction": 'max'},
                 group_args: dict = {'NAME': 'ballquery'}, sa_layers: int = 1, sa_use_res: bool = False,
                 **kwargs):
        super().__init__()
        if isinstance(block, str):
            block = _BLOCKS[block]
        self.blocks = blocks
        self.strides = strides
        self.in_channels = in_channels
        self.aggr_args = aggr_args
        self.norm_args = kwargs.get('norm_args', {'norm': 'bn'})
        self.act_args = kwargs.get('act_args', {'act': 'relu'})
        self.conv_args = kwargs.get('conv_args', None)
        self.sampler = kwargs.get('sampler', 'fps')
        self.expansion = kwargs.get('expansion', 4)
        self.sa_layers = sa_layers
        self.sa_use_res = sa_use_res
        self.use_res = kwargs.get('use_res', True)
        radius_scaling = kwargs.get('radius_scaling', 2)
        nsample_scaling = kwargs.get('nsample_scaling', 1)

        self.radii = self._to_full_list(radius, radius_scaling)
        self.nsample = self._to_full_list(nsample, nsample_scaling)
        logging.info(f'radius: {self.radii},\n nsample: {self.nsample}')

        channels = []  # width doubles at every strided stage
        for stride in strides:
            if stride != 1:
                width *= 2
            channels.append(width)

        stages = []
        for i in range(len(blocks)):
            # like the reference, the caller's group_args object is written to (pointnext_AA.py:362-363)
            group_args.radius = self.radii[i]
            group_args.nsample = self.nsample[i]
            stages.append(self._make_enc(block, channels[i], blocks[i], stride=strides[i], group_args=group_args,
                                         is_head=i == 0 and strides[i] == 1))
        self.encoder = nn.Sequential(*stages)
        self.out_channels = channels[-1]
        self.channel_list = channels

    def _to_full_list(self, param, param_scaling=1):
        """One value per block of every stage (pointnext_AA.py:374-392)."""
        out = []
        if isinstance(param, List):
            for i, value in enumerate(param):
                value = [value] if not isinstance(value, List) else value
                if len(value) != self.blocks[i]:
                    value += [value[-1]] * (self.blocks[i] - len(value))
                out.append(value)
        else:
            for i, stride in enumerate(self.strides):
                if stride == 1:
                    out.append([param] * self.blocks[i])
                else:
                    out.append([param] + [param * param_scaling] * (self.blocks[i] - 1))
                    param *= param_scaling
        return out

    def _make_enc(self, block, channels, blocks, stride, group_args, is_head=False):
        radii, nsample = group_args.radius, group_args.nsample
        group_args.radius, group_args.nsample = radii[0], nsample[0]
        layers = [SetAbstraction(self.in_channels, channels, self.sa_layers if not is_head else 1, stride,
                                 group_args=group_args, sampler=self.sampler, norm_args=self.norm_args,
                                 act_args=self.act_args, conv_args=self.conv_args, is_head=is_head,
                                 use_res=self.sa_use_res, **self.aggr_args)]
        self.in_channels = channels
        for i in range(1, blocks):
            group_args.radius, group_args.nsample = radii[i], nsample[i]
            layers.append(block(self.in_channels, aggr_args=self.aggr_args, norm_args=self.norm_args,
                                act_args=self.act_args, group_args=group_args, conv_args=self.conv_args,
                                expansion=self.expansion, use_res=self.use_res))
        return nn.Sequential(*layers)

    def forward_cls_feat(self, p0, f0=None):
        if hasattr(p0, 'keys'):
            p0, f0 = p0['pos'], p0.get('x', None)
        if f0 is None:
            f0 = p0.clone().transpose(1, 2).contiguous()
        for stage in self.encoder:
            p0, f0 = stage([p0, f0])
        return f0.squeeze(-1)

    @torch.no_grad()
    def plan_geometry(self, p0):
        """Coordinate-only work of the whole encoder: per stage a list with one plan per block.  Blocks of
        one stage that query the same cloud with the same radius / nsample (every InvResMLP of a stage,
        pointnext_AA.py:419-427) share one neighbour search."""
        plans, p = [], p0
        for stage in self.encoder:
            blocks, cache = [], {}
            for blk in stage:
                if isinstance(blk, SetAbstraction):
                    g = blk.plan(p)
                    p = g['new_p']
                else:
                    grouper = blk.convs.grouper
                    key = (getattr(grouper, 'radius', None), getattr(grouper, 'nsample', None))
                    if key not in cache:
                        cache[key] = blk.plan(p)
                    g = cache[key]
                blocks.append(g)
            plans.append(blocks)
        return plans

    def forward_seg_feat_ACE(self, p0, f0=None):
        """-> p[6], f[6], stageACE_list = {'inputs', 'down', 'up'}; 'down' and 'up' are the SAME list of
        per-stage dicts {p_out (B*n,3), f_out (B*n,C), offset int32 [B*n]} (pointnext_AA.py:439-467)."""
        stageACE_list = {'inputs': p0}
        geometry = None
        if hasattr(p0, 'keys'):
            geometry = p0.get('_geometry', None)
            p0, f0 = p0['pos'], p0.get('x', None)
        if f0 is None:
            f0 = p0.clone().transpose(1, 2).contiguous()
        if geometry is None:
            geometry = {'encoder': self.plan_geometry(p0)}
        stageACE_list['geometry'] = geometry
        p, f, down = [p0], [f0], []
        for i, stage in enumerate(self.encoder):
            pf = [p[-1], f[-1]]
            for blk, g in zip(stage, geometry['encoder'][i]):
                pf = blk(pf, geom=g)
            _p, _f = pf
            p.append(_p)
            f.append(_f)
            if i != len(self.encoder) - 1:
                flat_p = torch.flatten(_p, start_dim=0, end_dim=1)
                # the whole flattened batch is ONE segment: neighbours are searched across samples
                offset = _segment_offset(flat_p.shape[0], flat_p.device)
                down.append(_Stage(flat_p, _f, offset))  # 'f_out' = flatten(_f.transpose(1, 2)) on first access
        stageACE_list['down'] = down
        stageACE_list['up'] = down  # decoder overwrites ['f_out'] in place
        return p, f, stageACE_list

    def forward(self, p0, f0=None):
        return self.forward_seg_feat_ACE(p0, f0)


@MODELS.register_module()
class PointNextDecoder_AMContrast3D(nn.Module):
    def __init__(self, encoder_channel_list: List[int], decoder_layers: int = 2, decoder_stages: int = 4,
                 **kwargs):
        super().__init__()
        self.decoder_layers = decoder_layers
        self.in_channels = encoder_channel_list[-1]
        skip_channels = encoder_channel_list[:-1]
        if len(skip_channels) < decoder_stages:
            skip_channels.insert(0, kwargs.get('in_channels', 3))
        fp_channels = encoder_channel_list[:decoder_stages]
        n = len(fp_channels)
        stages = [None] * n
        for i in range(-1, -n - 1, -1):  # coarse to fine: in_channels chains through
            stages[i] = self._make_dec(skip_channels[i], fp_channels[i])
        self.decoder = nn.Sequential(*stages)
        self.out_channels = fp_channels[-n]

    def _make_dec(self, skip_channels, fp_channels):
        mlp = [skip_channels + self.in_channels] + [fp_channels] * self.decoder_layers
        self.in_channels = fp_channels
        return nn.Sequential(FeaturePropogation(mlp))

    @torch.no_grad()
    def plan_geometry(self, p):
        """3-NN indices / weights of every decoder level (coordinate-only), keyed by level -1 .. -n"""
        return {i: FeaturePropogation.plan(p[i - 1], p[i]) for i in range(-1, -len(self.decoder) - 1, -1)}

    def forward_then_ACE(self, p, f, stageACE_list):
        geometry = stageACE_list.get('geometry') if isinstance(stageACE_list, dict) else None
        if geometry is None:
            geometry = stageACE_list['geometry'] = {}
        if 'decoder' not in geometry:
            geometry['decoder'] = self.plan_geometry(p)
        for i in range(-1, -len(self.decoder) - 1, -1):
            f[i - 1] = self.decoder[i][1:](
                [p[i], self.decoder[i][0]([p[i - 1], f[i - 1]], [p[i], f[i]], geom=geometry['decoder'][i])])[1]
            # decoder embedding of this resolution, (B*n, C) rows, for the contrastive loss
            _set_embedding(stageACE_list['up'][i], f[i - 1])
        return f[-len(self.decoder) - 1].squeeze(-1), stageACE_list

    def forward(self, p, f, stageACE_list):
        return self.forward_then_ACE(p, f, stageACE_list)
