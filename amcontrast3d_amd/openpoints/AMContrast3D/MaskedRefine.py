"""Masked refinement of AMContrast3D++: points predicted to be highly ambiguous take the decoder feature of their
least ambiguous neighbour.

Drop-in for openpoints/AMContrast3D/MaskedRefine.py:7-131 (``RefinementMethod``), evaluated with the same tensor
operations in the same order.  Two things a reader might take for slips are part of the reference's behaviour and
are kept: the (B, D, n) feature tensor is REINTERPRETED as (B*n, D) rows by ``view`` (not transposed) before the
neighbour gather and the result is reinterpreted back (:64, :106), and the k-NN runs over all B*n points as one
segment (:61-62), so neighbours may come from other clouds of the batch.
"""
import torch

from openpoints.cpp.pointops.functions import pointops

RATE_ON_DEVICE = False  # True: refine rates are returned as device tensors (no host read-back per stage)


class RefinementMethod:
    def __init__(self, stage_list, p, f, a, i, B, K, fusion, threshold_max, threshold, gamma):
        self.stage_list = stage_list
        self.position, self.feature, self.ambiguity = p, f, a
        self.i, self.batch, self.sample_k = i, B, K
        self.fusion, self.threshold_max, self.threshold, self.gamma = fusion, threshold_max, threshold, gamma

    # ---- variants on the predicted ambiguity map (APM linear_mapping=True) ------------------------------
    def _a_map(self):
        am = self.stage_list['ambiguity_map'][self.i]
        return am.unsqueeze(0).view(self.batch, am.shape[1], -1)  # [N, D] reinterpreted as [b, D, n] (:24)

    def MapAttention(self):
        raise NotImplementedError("cross-attention refinement (APM/attention.py) is outside this build; "
                                  "the shipped configs use cross_attention: False")

    def MapSum(self):
        self.feature = self.feature + self._a_map()
        return self.feature

    def MapMultiply(self):
        self.feature = torch.mul(self.feature, self._a_map())
        return self.feature

    def Multiply(self):
        self.feature = torch.mul(self.feature, self.ambiguity)
        return self.feature

    # ---- the default: self mask x cross mask --------------------------------------------------------------
    def DualMasks(self):
        geometry = self.stage_list.get('geometry') if hasattr(self.stage_list, 'get') else None
        planned = geometry.get('refine') if geometry is not None else None
        if planned is not None and self.i in planned and planned[self.i].shape[1] == self.sample_k - 1:
            neighbor_idx = planned[self.i]      # coordinate-only: prepared with the rest of the geometry plan
        else:
            xyz = self.position.view(-1, 3)
            from openpoints.models.backbone.pointnext_AA import _segment_offset
            o = _segment_offset(xyz.shape[0], xyz.device)  # IntTensor([b*n]).cuda() of the reference, cached
            neighbor_idx, _ = pointops.knnquery(self.sample_k, xyz, xyz, o, o)  # (b*n, K), one segment
            neighbor_idx = neighbor_idx[..., 1:].contiguous()
        D = self.feature.shape[1]
        import os
        if (self.fusion == 'MIN' and self.feature.is_cuda and self.feature.dtype == torch.float32
                and self.ambiguity.dtype == torch.float32 and neighbor_idx.dtype == torch.int32 and self.feature.dim() == 3
                and not os.environ.get("AMC3D_NO_FUSED_REFINE")):
            # the whole rule as two kernels (csrc/refine.hip): same values, same reinterpretations
            from amcontrast3d_amd import ops
            self.sample_k -= 1
            self.feature, count = ops.MaskedRefineDual.apply(self.feature, self.ambiguity, neighbor_idx, self.threshold,
                                                             self.threshold_max, self.gamma)
            if RATE_ON_DEVICE or torch.cuda.is_current_stream_capturing():
                return self.feature, count.to(torch.float32) / self.ambiguity.numel() * 100
            return self.feature, (count.item() / self.ambiguity.numel()) * 100
        f_rows = self.feature.view(-1, D)      # memory reinterpretation, see the module docstring
        a_rows = self.ambiguity.view(-1, 1)
        self.sample_k -= 1                      # drop the self match
        m = neighbor_idx.shape[0]
        flat = neighbor_idx.view(-1).long()
        neighbor_ambiguity = a_rows.index_select(0, flat).view(m, self.sample_k, 1)
        if self.fusion == 'MIN':
            # The reference gathers all K-1 neighbour rows (m, K-1, D), multiplies them by a one-hot of the arg-min
            # ambiguity and sums over the neighbours (:103-113): that IS the arg-min neighbour's row (plus exact
            # zeros).  Gathering only that row moves 1/(K-1) of the bytes, forward and backward (index_add).
            good_idx = torch.min(neighbor_ambiguity, 1).indices                            # (m, 1), first on ties
            best = neighbor_idx.long().gather(1, good_idx).view(-1)
            cross = f_rows.index_select(0, best).view(self.feature.shape[0], D, -1)
        else:
            neighbor_feature = f_rows.index_select(0, flat).view(m, self.sample_k, D)
            cross = self.cross_mask(neighbor_ambiguity, neighbor_feature, D)
        self_mask, rate = self.self_mask()
        f_new = self.feature * ~self_mask + cross * self_mask
        self.feature = self.gamma * f_new + (1 - self.gamma) * self.feature  # constant updating rate
        return self.feature, rate

    def cross_mask(self, neighbor_ambiguity, neighbor_feature, D):
        if self.fusion == 'MIN':
            # feature row of the neighbour with the smallest predicted ambiguity (first one on ties: torch.min)
            good_idx = torch.min(neighbor_ambiguity, 1).indices                       # (m, 1)
            good = torch.gather(neighbor_feature, 1, good_idx.unsqueeze(-1).expand(-1, 1, D)).squeeze(1)
            # the reference builds a one-hot (m, K, D) mask, multiplies and sums over K: the same row, plus
            # (K-1) exact zeros -- identical values
        elif self.fusion == 'MIN_ALL0':
            good = torch.mean(neighbor_feature * ~neighbor_ambiguity.gt(0), dim=1)
        else:
            raise ValueError(f"unknown fusion {self.fusion!r}")
        return good.view(self.feature.shape[0], D, -1)

    def self_mask(self):
        mask = self.ambiguity.le(self.threshold_max) * self.ambiguity.ge(self.threshold)
        count = torch.count_nonzero(mask.long())
        if mask.is_cuda and (RATE_ON_DEVICE or torch.cuda.is_current_stream_capturing()):
            # under hipGraph capture, or when the caller asked for it (amcontrast3d_amd.train keeps the GPU running
            # ahead), the percentage stays a 0-dim tensor: the reference's .item() is a host sync per stage
            return mask, count.to(torch.float32) / self.ambiguity.numel() * 100
        return mask, (count.item() / self.ambiguity.numel()) * 100
