"""Adaptive-margin contrastive head over per-stage point embeddings.

Drop-in for openpoints/AMContrast3D/MarginContrast.py:
    AmbiguityHead   :15-52    per-stage ambiguity a_i only (used by the ++ variant)
    ContrastHead    :56-273   per stage: 24-NN (self dropped) -> neighbour labels ->
                              positive mask -> a_i -> cosine similarity -> soft-NN loss
                              with margin m_i = mu * a_i + nu on the positive pairs

Only the configuration the shipped configs select is implemented
(dist_cos + contrast_softnn_margin, cfgs/*/AMContrast3D-AA.yaml:6-30); the other
similarity / loss variants of the reference are unreachable from its configs.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from openpoints.cpp.pointops.functions import pointops
from .AEF.ambiguity import ambiguity_function
from .AEF.function import _eps
from .AEF.utils import fetch_pxo, get_ftype, get_subscene_class, get_subscene_label_CBL  # noqa: F401


def _posmask_cnt(labels, neighbor_label):
    """same arg-max class as the anchor -> (m, k) bool (MarginContrast.py:111-115)"""
    return torch.argmax(torch.unsqueeze(labels, -2), -1) == torch.argmax(neighbor_label, -1)


def plan_stage(n, i, stageACE_list, target, nstride, num_classes, ignore_index, ambiguity_args, ftype='f_out'):
    """Everything of one loss stage that depends only on coordinates and labels (not on the
    embeddings): the stage's k-NN (self match dropped, a strided view), the positive mask and the
    ambiguity a_i -- MarginContrast.py:220-240.  Cached in stageACE_list['geometry']['loss'] when a
    caller prepared it ahead of time (amcontrast3d_amd.geometry.precompute)."""
    from amcontrast3d_amd import ops
    p, o = stageACE_list[n][i]['p_out'], stageACE_list[n][i]['offset']  # no embeddings needed here
    labels, _ = get_subscene_class(n, i, stageACE_list, target, nstride, num_classes, ignore_index)
    knn = ops.knnquery_squared if p.is_cuda else pointops.knnquery  # (distances: any monotone image serves contrast_mutual)
    neighbor_idx, neighbor_d2 = knn(ambiguity_args.nsample, p, p, o, o)
    neighbor_idx = neighbor_idx[..., 1:]  # drop the self match: a strided view, no copy
    posmask = ops.posmask_from_labels(labels, neighbor_idx)
    a, shares = ambiguity_function(p, posmask, neighbor_idx.shape[1], neighbor_idx, ambiguity_args.cctype,
                                   ambiguity_args.ccbeta, ambiguity_args.vis, ambiguity_args.nu)
    # the anchors the loss keeps (0 < a <= 1, :250-252) as a compact list for the fused contrast kernels
    anchors = ops.select_anchors(a) if a.is_cuda and a.dtype == torch.float32 else None
    # Reverse structure for the loss backward, so that it GATHERS every gradient row instead of scattering rows with float
    # atomics (0.71 ms per step at the chip's float-atomic rate in round 2).  Default (round 3): the mutual-edge form --
    # ~90 % of the k-NN edges are mutual and need no list at all (ops.contrast_mutual: a mutual bit per edge + reverse lists
    # of the remaining tenth).  AMC3D_CONTRAST_CSR=1: round 2's reverse lists of ALL edges (0.79 ms of integer atomics on
    # the geometry stream); AMC3D_CONTRAST_ATOMIC=1: the float-atomic form.
    rev = mutual = None
    if anchors is not None and os.environ.get("AMC3D_CONTRAST_CSR"):
        rev = ops.contrast_csr(neighbor_idx, anchors)
    elif anchors is not None and neighbor_idx.shape[1] <= 64 and not os.environ.get("AMC3D_CONTRAST_ATOMIC"):
        # (one segment of more than k + 1 points: every list is full, membership follows from one distance comparison)
        d2 = neighbor_d2[..., 1:] if (torch.is_tensor(neighbor_d2) and neighbor_d2.dtype == torch.float32 and o.numel() == 1
                                      and neighbor_d2.shape[0] == neighbor_idx.shape[0] > neighbor_idx.shape[1] + 2) else None
        mutual, rev = ops.contrast_mutual(neighbor_idx, a, d2)
    return {'neighbor_idx': neighbor_idx, 'posmask': posmask, 'ambiguity': a, 'shares': shares, 'anchors': anchors,
            'rev': rev, 'mutual': mutual}


def _stage_plan(n, i, stageACE_list, target, nstride, num_classes, ignore_index, ambiguity_args, ftype):
    geometry = stageACE_list.get('geometry') if hasattr(stageACE_list, 'get') else None
    if geometry is not None and 'loss' in geometry:
        return geometry['loss'][i]
    return plan_stage(n, i, stageACE_list, target, nstride, num_classes, ignore_index, ambiguity_args, ftype)


class AmbiguityHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.nstride = torch.tensor([4, 4, 4, 4])
        self.ftype = get_ftype('latent')[0]
        self.posmask_func = _posmask_cnt
        self.main = self.point_ambiguity

    def posmask_cnt(self, labels, neighbor_label):
        return _posmask_cnt(labels, neighbor_label)

    def point_ambiguity(self, n, i, stageACE_list, target, num_classes, ignore_index, ambiguity_args):
        return _stage_plan(n, i, stageACE_list, target, self.nstride, num_classes, ignore_index, ambiguity_args,
                           self.ftype)['ambiguity']

    def forward(self, target, stageACE_list, num_classes, ignore_index, ambiguity_args):
        return [self.main(ambiguity_args.stages, i, stageACE_list, target, num_classes, ignore_index, ambiguity_args)
                for i in range(ambiguity_args.stages_num)]


class ContrastHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.nstride = torch.tensor([4, 4, 4, 4])
        self.stages = [('up', 0), ('up', 1), ('up', 2), ('up', 3)]
        self.ftype = get_ftype('latent')[0]
        self.project = None
        self.dist_func = self.dist_cos
        self.contrast_func = self.contrast_softnn_margin
        self.posmask_func = self.posmask_cnt
        self.main_contrast = self.point_contrast_margin

    def dist_cos(self, features, neighbor_feature):
        """(m,C), (m,k,C) -> cosine similarity (m,k)"""
        return F.cosine_similarity(torch.unsqueeze(features, -2), neighbor_feature, dim=2)

    def posmask_cnt(self, labels, neighbor_label):
        return _posmask_cnt(labels, neighbor_label)

    def contrast_softnn_margin(self, dist, posmask, ambiguity, ambiguity_args, invalid_mask=None):
        """-log( sum_+ e^{s'} / sum e^{s'} + eps ), s' = (s - m_i)/T on positives, s/T on negatives
        (MarginContrast.py:117-174)."""
        if ambiguity_args.margin == 'constant':
            margin = ambiguity_args.nu
        elif ambiguity_args.margin == 'adaptive':
            margin = ambiguity_args.mu * torch.unsqueeze(ambiguity, -1) + ambiguity_args.nu
        elif ambiguity_args.margin == 'learned':
            u = torch.mean(dist * ~posmask, 1)
            v = torch.mean(dist * posmask, 1)
            margin = (torch.unsqueeze(u, -1) - 1) * torch.unsqueeze(ambiguity, -1) + torch.unsqueeze(v, -1)

        if ambiguity_args.db == '-m':
            dist = (dist - margin) * posmask + dist * ~posmask
        elif ambiguity_args.db == '+m':
            dist = dist * posmask + (dist + margin) * ~posmask
        else:
            dist = dist * posmask + dist * ~posmask

        if ambiguity_args.temperature is not None:
            dist = dist / ambiguity_args.temperature
        exp = torch.exp(dist)
        if invalid_mask is not None:
            exp = exp * (1 - invalid_mask)

        pos = torch.sum(exp * posmask, axis=-1)
        neg = torch.sum(exp * (1 - posmask.int()), axis=-1)
        pos_neg = torch.sum(exp, axis=-1)
        if ambiguity_args.supervisedCL == 'Method1':
            loss = pos / pos_neg + _eps
        elif ambiguity_args.supervisedCL == 'Method2':
            loss = (exp * posmask) / (exp * posmask + neg.unsqueeze(-1)) + _eps
            loss = torch.sum(loss, axis=-1) / (torch.sum(posmask.int(), axis=-1) + _eps)
        return -torch.log(loss)

    def plan(self, target, stageACE_list, num_classes, ignore_index, ambiguity_args):
        """coordinate / label-only part of every stage (see plan_stage)"""
        from amcontrast3d_amd import ops
        # the full-resolution cloud is searched four times (its own neighbours, then the label votes of the three
        # coarser stages): one cell grid serves all four
        with torch.no_grad(), ops.knn_grid_reuse():
            return [plan_stage(ambiguity_args.stages, i, stageACE_list, target.flatten(), self.nstride, num_classes,
                               ignore_index, ambiguity_args, self.ftype) for i in range(ambiguity_args.stages_num)]

    def point_contrast_margin(self, n, i, stageACE_list, target, num_classes, ignore_index, ambiguity_args):
        from amcontrast3d_amd import ops
        g = _stage_plan(n, i, stageACE_list, target, self.nstride, num_classes, ignore_index, ambiguity_args,
                        self.ftype)
        neighbor_idx, posmask, ambiguity_soft = g['neighbor_idx'], g['posmask'], g['ambiguity']
        target_ai = ambiguity_soft
        output_ai = stageACE_list['ambiguity'][i].flatten() if 'ambiguity' in stageACE_list.keys() else None
        fused = (ambiguity_args.margin == 'adaptive' and ambiguity_args.db == '-m'
                 and ambiguity_args.supervisedCL == 'Method1' and ambiguity_args.temperature is not None)
        # the decoder's channel-major embedding, where the stage entry still holds it: the fused stage reads it as it is and
        # the (B*n, C) copy of pointnext_AA.py:518-519 is not made
        stage = stageACE_list[n][i]
        f_cm = stage.channel_major() if fused and hasattr(stage, 'channel_major') else None
        if (f_cm is not None and not os.environ.get("AMC3D_LOSS_ROWS")
                and ops.contrast_stage_supported_cm(f_cm, g.get('anchors'), g.get('rev'), g.get('mutual'))):
            loss = ops.contrast_stage_cm(f_cm, neighbor_idx, posmask, ambiguity_soft, ambiguity_args.mu, ambiguity_args.nu,
                                         ambiguity_args.temperature, g['anchors'], g['rev'], g['mutual'])
            return loss, output_ai, target_ai
        features = fetch_pxo(n, i, stageACE_list, self.ftype)[1]
        if features.dtype != torch.float32:  # embeddings produced under autocast (use_amp): the loss is evaluated in fp32
            features = features.float()
        k = neighbor_idx.shape[1]
        if fused:
            # anchors with 0 < a <= 1 enter the loss (MarginContrast.py:250-257); selection, cosine
            # similarity, margin soft-NN loss and the mean are one forward and one backward kernel
            loss = ops.contrast_stage(features, neighbor_idx, posmask, ambiguity_soft, ambiguity_args.mu,
                                      ambiguity_args.nu, ambiguity_args.temperature, g.get('anchors'), g.get('rev'),
                                      g.get('mutual'))
            return loss, output_ai, target_ai
        # other margin / decision-boundary / Method2 variants: composed from the torch-level pieces
        keep = torch.logical_and(0 < ambiguity_soft, ambiguity_soft <= 1)
        m = neighbor_idx.shape[0]
        neighbor_feature = features[neighbor_idx.reshape(-1).long(), :].view(m, k, features.shape[1])
        dist = self.dist_func(features[keep], neighbor_feature[keep])
        loss = self.contrast_func(dist, posmask[keep], ambiguity_soft[keep], ambiguity_args)
        return torch.mean(loss), output_ai, target_ai

    def forward(self, output, target, stageACE_list, num_classes, ignore_index, ambiguity_args):
        loss_sum = 0
        target_ai_list = []
        for i in range(ambiguity_args.stages_num):
            loss, _, target_ai = self.main_contrast(ambiguity_args.stages, i, stageACE_list, target, num_classes,
                                                    ignore_index, ambiguity_args)
            loss_sum += loss
            target_ai_list.append(target_ai)
        return loss_sum, torch.cat(target_ai_list), target_ai_list
