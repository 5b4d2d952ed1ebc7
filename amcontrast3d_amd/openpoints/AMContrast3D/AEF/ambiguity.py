"""Ambiguity estimation function a_i (AMContrast3D/AEF/ambiguity.py:11-93).

Same inputs, outputs and arithmetic as the reference; the per-boundary-point
Python loop that builds the neighbour-coordinate tensor with a growing
``torch.cat`` (ambiguity.py:32-35, O(m_b^2) bytes and 2*m_b launches) and the
torch ops around it are two gfx950 kernels here, and the five bucket percentages
-- five ``.item()`` host syncs in the reference (:79-91), unused by the loss --
are evaluated lazily.
"""
import torch


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
    """p (m,3), posmask (m,k) bool, neighbor_idx (m,k) int32 -> a (m) in [0,1], bucket shares.

    n+ = #same-class neighbours; a = |n+ - max n+| / max n+  (0 inner, 1 isolated);
    for 0 < n+ < max:  a = 1 / (1 + e^(beta (n+/d+ - n-/d-)))  with d+- the summed
    (squared, Method2) distances to the positive / negative neighbours.  One fused kernel pair
    (amc3d_ambiguity) evaluates it with the reference's fp32 operation order, including the
    expanded -2ab + a^2 + b^2 distance of AEF/function.py:18-39."""
    if ambiguity_vis:
        raise NotImplementedError("ambiguity_args.vis needs the reference's pyvista viewer (AMContrast3D/vis.py)")
    if ambiguity_type not in ('Method1', 'Method2', 'Method3'):
        raise ValueError(f'unknown cctype {ambiguity_type}')
    from amcontrast3d_amd import ops
    a = ops.ambiguity(p.contiguous(), posmask.contiguous(), neighbor_idx, ambiguity_type, ambiguity_beta)
    return a, _LazyShares(a, nu)
