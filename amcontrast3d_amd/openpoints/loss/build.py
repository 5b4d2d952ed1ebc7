"""LOSS registry and the cross-entropy + adaptive-margin-contrast criterion.

Drop-in for openpoints/loss/build.py: ``LOSS`` (:9-12), ``CrossEntropyAce`` (:324-346),
``build_criterion_from_cfg`` (:348-357).
"""
import torch
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss

from openpoints.AMContrast3D.MarginContrast import ContrastHead
from openpoints.utils import registry

LOSS = registry.Registry('loss')
LOSS.register_module(name='CrossEntropy', module=CrossEntropyLoss)
LOSS.register_module(name='CrossEntropyLoss', module=CrossEntropyLoss)
LOSS.register_module(name='BCEWithLogitsLoss', module=BCEWithLogitsLoss)


def _cross_entropy(crit, logit, target):
    """(CE value, flattened target): nn.CrossEntropyLoss on the (B*N, ncls) view of the logits, through the fused
    kernel when the module is the plain default"""
    if (type(crit) is CrossEntropyLoss and crit.weight is None and crit.label_smoothing == 0.0
            and crit.reduction == 'mean' and logit.is_cuda and logit.dtype == torch.float32 and logit.dim() == 3):
        # same value without the (B*N, ncls) transposed copy: one fused pass over the (B,ncls,N) logits
        from amcontrast3d_amd.ops import cross_entropy_mean
        return cross_entropy_mean(logit, target, crit.ignore_index), target.flatten()
    target = target.flatten()
    return crit(logit.transpose(1, 2).reshape(-1, logit.shape[1]), target), target


def _l1_mean(crit, pred, target, rows=1024):
    """nn.L1Loss()(pred, target) in two small reduction stages on the GPU.  torch reduces a long vector to one number with a
    multi-block kernel whose arrival counters it zeroes with cudaMemsetAsync; recorded into a graph that is a memset NODE, and
    ROCm 7.2 does not reliably order a memset node before the kernel nodes behind it at replay (the fault
    the radix sort's memsets caused) -- here the counters of a reduction, i.e. a sum that may silently come out wrong.  Rows
    of `rows` elements summed per row, then the row sums: neither stage crosses workgroups.  Same value up to the order of the
    fp32 additions (1e-7 relative); anything but the plain mean on CUDA tensors goes to the module itself."""
    if not (type(crit) is torch.nn.L1Loss and crit.reduction == 'mean' and pred.is_cuda and pred.dim() == 1
            and pred.shape == target.shape and pred.numel() > rows):
        return crit(pred, target)
    d = (pred - target).abs()
    n = d.numel()
    pad = (-n) % rows
    if pad:
        d = torch.nn.functional.pad(d, (0, pad))
    return d.view(-1, rows).sum(1).sum() / n


@LOSS.register_module()
class CrossEntropyAce(torch.nn.Module):
    """w1 * CE(logits, target) + w2 * sum_stage contrast(stage).  Like the reference, the
    constructor accepts and ignores label_smoothing / weight / ignore_index (build.py:326-329)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.creterion = CrossEntropyLoss()  # attribute name as in the reference
        self.contrast_head = ContrastHead()

    def forward(self, logit, target, stageACE_list, num_classes, ignore_index, ambiguity_args):
        ce, target = _cross_entropy(self.creterion, logit, target)
        contrast, _, _ = self.contrast_head(logit, target, stageACE_list, num_classes, ignore_index, ambiguity_args)
        return ambiguity_args.w1 * ce + ambiguity_args.w2 * contrast


@LOSS.register_module()
class CrossEntropyAcePre(torch.nn.Module):
    """AMContrast3D++ objective (loss/build.py:281-319): w1 * CE + w2 * contrast for the segmentation, and
    w3 * L1(predicted ambiguity, AEF ambiguity) for the APM, returned separately:
    (segmentation loss, w1*CE, w2*contrast, w3*regression)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.creterion = CrossEntropyLoss()
        self.contrast_head = ContrastHead()
        self.MAE = torch.nn.L1Loss()
        self.MSE = torch.nn.MSELoss()
        self.HUBER = torch.nn.HuberLoss(reduction='mean', delta=0.1)

    def forward(self, logit, target, stageACE_list, num_classes, ignore_index, ambiguity_args):
        ce, target = _cross_entropy(self.creterion, logit, target)
        contrast, target_ai, _ = self.contrast_head(logit, target, stageACE_list, num_classes, ignore_index,
                                                    ambiguity_args)
        logits_ai = torch.cat(stageACE_list['ambiguity']).flatten()
        regression = _l1_mean(self.MAE, logits_ai, target_ai)
        ce = ambiguity_args.w1 * ce
        contrast = ambiguity_args.w2 * contrast
        regression = ambiguity_args.w3 * regression
        return ce + contrast, ce, contrast, regression


def build_criterion_from_cfg(cfg, **kwargs):
    """Build the criterion named by ``cfg.NAME``."""
    return LOSS.build(cfg, **kwargs)
