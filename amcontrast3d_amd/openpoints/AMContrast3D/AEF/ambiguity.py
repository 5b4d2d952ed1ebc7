"""Ambiguity estimation function a_i (AMContrast3D/AEF/ambiguity.py:11-93).

Same inputs, outputs and arithmetic as the reference; the per-boundary-point
Python loop that builds the neighbour-coordinate tensor with a growing
``torch.cat`` (ambiguity.py:32-35, O(m_b^2) bytes and 2*m_b launches) is one
indexed gather here, and the five bucket percentages -- five ``.item()`` host
syncs in the reference (:79-91), unused by the loss -- are evaluated lazily.
"""
import math

import torch

from .function import _eps, inverse_sigmoid_function, square_distance


class _LazyShares:
    """count_low_semi_high: [%a==0, %low, %semi, %high, %a==1]; syncs the device only if read."""

    def __init__(self, a, nu):
        self._a, self._nu, self._v = a, nu, None

    def _values(self):
        if self._v is None:
            a, nu_m = self._a, self._nu * 10
            c = torch.ceil(a * 10)
            masks = torch.stack([a == 0, (0 < c) & (c < nu_m), c == nu_m, (nu_m < c) & (c < 10), c == 10])
            n = len(a)
            self._v = [round(s / n * 100, 2) for s in masks.sum(1).tolist()]
        return self._v

    def __iter__(self):
        return iter(self._values())

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return self._values()[i]

    def __repr__(self):
        return repr(self._values())


def ambiguity_function(p, posmask, nsample, neighbor_idx, ambiguity_type, ambiguity_beta, ambiguity_vis, nu):
    """p (m,3), posmask (m,k) bool, neighbor_idx (m,k) -> a (m) in [0,1], bucket shares.

    n+ = #same-class neighbours; a = |n+ - max n+| / max n+  (0 inner, 1 isolated);
    for 0 < n+ < max:  a = 1 / (1 + e^(beta (n+/d+ - n-/d-)))  with d+- the summed
    (squared, Method2) distances to the positive / negative neighbours."""
    mask_num = torch.sum(posmask.int(), -1)
    top = torch.max(mask_num)
    a = torch.abs(mask_num.add(-top)).div(top)
    boundary = torch.logical_and(0 < mask_num, mask_num < top)  # == (0 < a) & (a < 1)

    mask_b = posmask[boundary]
    n_pos = torch.sum(mask_b.int(), -1)
    n_neg = torch.sum(1 - mask_b.int(), -1)

    if ambiguity_type == 'Method1':
        d_pos = torch.full(n_pos.shape, 5.0, device=p.device)
        d_neg = torch.full(n_neg.shape, 5.0, device=p.device)
    elif ambiguity_type in ('Method2', 'Method3'):
        centre = p[boundary].unsqueeze(1)            # (m_b,1,3)
        nbrs = p[neighbor_idx[boundary].long()]      # (m_b,k,3)
        dd = square_distance(centre, nbrs).squeeze(1)
        if ambiguity_type == 'Method3':
            dd = torch.sqrt(torch.abs(dd) + _eps)
        d_pos = torch.sum(mask_b.int() * dd, -1)
        d_neg = torch.sum((1 - mask_b.int()) * dd, -1)
    else:
        raise ValueError(f'unknown cctype {ambiguity_type}')

    cc_pos = n_pos / d_pos
    cc_neg = n_neg / d_neg
    t = torch.full(cc_pos.shape, math.e, device=p.device)
    ai_soft = inverse_sigmoid_function(cc_pos - cc_neg, t, ambiguity_beta)

    if ambiguity_vis:
        raise NotImplementedError("ambiguity_args.vis needs the reference's pyvista viewer (AMContrast3D/vis.py)")
    a[boundary] = ai_soft
    return a, _LazyShares(a, nu)
