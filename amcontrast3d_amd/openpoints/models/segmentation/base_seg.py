"""Segmentation wrapper and head.

Drop-in for openpoints/models/segmentation/base_seg.py:
    BaseSeg_M_AMContrast3D :17-94    encoder -> APM (predicted ambiguity per resolution) -> decoder with masked
                                     refinement -> head, returns (logits, stageACE_list, refine rate)
    BaseSeg_AMContrast3D   :97-126   encoder -> decoder -> head, returns (logits, stageACE_list)
    SegHead                :207-267  Conv1d+BN+ReLU(+Dropout) ... Conv1d, optional global max/avg concat
"""
import copy
import logging
from typing import List

import torch
import torch.nn as nn

from ..build import MODELS, build_model_from_cfg
from ..layers import create_convblock1d, run_convblocks


def _build_enc_dec(self, encoder_args, decoder_args):
    self.encoder = build_model_from_cfg(encoder_args)
    if decoder_args is not None:
        merged = copy.deepcopy(encoder_args)  # decoder sees the encoder's kwargs too (base_seg.py:22-26, 103-106)
        merged.update(decoder_args)
        merged.encoder_channel_list = self.encoder.channel_list if hasattr(self.encoder, 'channel_list') else None
        self.decoder = build_model_from_cfg(merged)
    else:
        self.decoder = None


def _build_head(self, cls_args):
    if cls_args is not None:
        if hasattr(self.decoder, 'out_channels'):
            in_channels = self.decoder.out_channels
        elif hasattr(self.encoder, 'out_channels'):
            in_channels = self.encoder.out_channels
        else:
            in_channels = cls_args.get('in_channels', None)
        cls_args.in_channels = in_channels  # written back into the caller's cfg, as the reference does
        self.head = build_model_from_cfg(cls_args)
    else:
        self.head = None


@MODELS.register_module()
class BaseSeg_M_AMContrast3D(nn.Module):
    def __init__(self, AEF_args=None, APM_args=None, encoder_args=None, decoder_args=None, cls_args=None, **kwargs):
        super().__init__()
        _build_enc_dec(self, encoder_args, decoder_args)  # module order = the reference's: encoder, decoder, APM, head
        if AEF_args is not None:
            self.AEF_args = AEF_args
        if APM_args is not None:
            self.APM = build_model_from_cfg(APM_args)
            self.name = APM_args.NAME
            self.linear_mapping = APM_args.linear_mapping
            self.cross_attention = APM_args.cross_attention
            self.feat_concate = APM_args.feat_concate
            self.nsample_k = APM_args.nsample_k
            self.threshold = APM_args.threshold
            self.threshold_max = APM_args.threshold_max
            self.gamma = APM_args.gamma
            self.fusion = APM_args.fusion
        _build_head(self, cls_args)
        if cls_args is not None:
            self.num_classes = cls_args.num_classes
            self.ignore_index = cls_args.ignore_index

    def forward(self, data):
        p, f, stageACE_list = self.encoder.forward(data)
        if self.name in ('APM_p', 'APM_p_Group', 'APM_p_Graph', 'APM_pp_SelfAtt'):
            # position-only predictors (base_seg.py:62-68): one ambiguity per point of p[1..4], no mapped embedding
            stageACE_list['ambiguity'] = [self.APM.forward(p[i]) for i in range(1, len(p) - 1)]
        elif self.name in ('APM_pf_ConCate', 'APM_pf_CrossAtt'):
            a, a_map = [], []  # (B*n, 1) ambiguities [and (B*n, D) maps] of the four resolutions p[1..4]
            for i in range(1, len(p) - 1):
                if self.linear_mapping:
                    r1, r2 = self.APM.forward(p[i], f[i])
                    a.append(r1)
                    a_map.append(r2)
                else:
                    a.append(self.APM.forward(p[i], f[i]))
            stageACE_list['ambiguity'] = a
            stageACE_list['ambiguity_map'] = a_map
        # any other name: no 'ambiguity' entry, and the decoder fails on the missing key as the reference's does
        f, stageACE_list, refine = self.decoder.forward(p, f, stageACE_list, self.linear_mapping, self.cross_attention,
                                                        self.feat_concate, self.nsample_k, self.threshold,
                                                        self.threshold_max, self.gamma, self.fusion, self.num_classes,
                                                        self.ignore_index, self.AEF_args)
        return self.head(f), stageACE_list, refine


@MODELS.register_module()
class BaseSeg_AMContrast3D(nn.Module):
    def __init__(self, encoder_args=None, decoder_args=None, cls_args=None, **kwargs):
        super().__init__()
        _build_enc_dec(self, encoder_args, decoder_args)
        _build_head(self, cls_args)

    def forward(self, data):
        p, f, stageACE_list = self.encoder.forward(data)
        f, stageACE_list = self.decoder.forward(p, f, stageACE_list)
        return self.head(f), stageACE_list


@MODELS.register_module()
class SegHead(nn.Module):
    def __init__(self, num_classes, in_channels, mlps=None, norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'},
                 dropout=0.5, global_feat=None, **kwargs):
        super().__init__()
        if kwargs:
            logging.warning(f"kwargs: {kwargs} are not used in {__class__.__name__}")
        if global_feat is not None:
            self.global_feat = global_feat.split(',')
            in_channels *= len(self.global_feat) + 1
        else:
            self.global_feat = None

        if mlps is None:
            mlps = [in_channels, in_channels] + [num_classes]
        else:
            if not isinstance(mlps, List):
                mlps = [mlps]
            mlps = [in_channels] + mlps + [num_classes]
        heads = []
        for i in range(len(mlps) - 2):
            heads.append(create_convblock1d(mlps[i], mlps[i + 1], norm_args=norm_args, act_args=act_args))
            if dropout:
                heads.append(nn.Dropout(dropout))
        heads.append(create_convblock1d(mlps[-2], mlps[-1], act_args=None))
        self.head = nn.Sequential(*heads)

    def forward(self, end_points):
        if self.global_feat is not None:
            g = []
            for kind in self.global_feat:
                if 'max' in kind:
                    g.append(torch.max(end_points, dim=-1, keepdim=True)[0])
                elif kind in ['avg', 'mean']:
                    g.append(torch.mean(end_points, dim=-1, keepdim=True))
            g = torch.cat(g, dim=1).expand(-1, -1, end_points.shape[-1])
            end_points = torch.cat((end_points, g), dim=1)
        return run_convblocks(self.head, end_points)
