"""Boundary / inner split of the evaluation loops (drop-in for openpoints/AMContrast3D/metrics.py:160-184).

`posmask_searching` marks, for every point, which of its nsample-1 nearest neighbours carry its own label; the
evaluation loops call a point "boundary" when 0 < #same < nsample (examples/segmentation/main_AA.py:470-476).
The reference builds one-hot labels, gathers an (m, nsample-1, classes) tensor and arg-maxes both sides; comparing
the integer labels directly (ops.posmask_from_labels, csrc/loss.hip) is the same predicate.
"""
import torch

from openpoints.cpp.pointops.functions import pointops


def posmask_searching(xyz, target, nsample, num_classes, ignore_index):
    """xyz (m,3) fp32, target (m) int64 -> posmask (m, nsample-1) bool, neighbor_idx (m, nsample-1) int32.
    ignore_index (ScanNet: -100) counts as one extra class, so unlabeled neighbours match unlabeled anchors."""
    from amcontrast3d_amd import ops
    target = target.reshape(-1)
    if ignore_index is not None:
        target = torch.where(target == ignore_index, num_classes, target)
    xyz = xyz.contiguous()
    o = torch.tensor([xyz.shape[0]], dtype=torch.int32, device=xyz.device)
    neighbor_idx, _ = pointops.knnquery(nsample, xyz, xyz, o, o)
    neighbor_idx = neighbor_idx[..., 1:].contiguous()  # drop the self match
    posmask = ops.posmask_from_labels(target.to(torch.int32).contiguous(), neighbor_idx)
    return posmask, neighbor_idx


@torch.no_grad()
def ambiguity_metrics(p, label, pred, posmask_test, nsample_test, neighbor_idx_test, cctype, ccbeta, vis,
                      cm_0, cm_low, cm_semi, cm_high, cm_1, nu):
    """Accuracy of a prediction per ambiguity level (AMContrast3D/metrics.py:33-157; called per test cloud when
    ambiguity_args.action is set, examples/segmentation/main_AA.py:686-702).

    a_i of every point (the fused kernels of AEF.ambiguity_function) is binned by floor(10 a + 1) in 1..11; five
    groups -- a = 0, low (< nu), semi (the nu bin), high, a = 1 -- get a confusion matrix each (the caller's
    cm_0 .. cm_1 are updated, as in the reference).  Returns, like the reference:
        ambiguity_soft (m), ratio {bin: accuracy}, ambiguity_count [5 shares in %], ratio_low_semi_high (the constant
        [1.0] * 5 the reference returns), cls {class: [5 shares in % of that class]}, and the rounded mIoU / mAcc / OA
        lists plus per-class counts of the five matrices.
    The reference prints these as it goes; this returns them only.  Its per-class thresholds are the literal bins 6
    and 11 (S3DIS nu = 0.5) regardless of `nu`; kept."""
    from openpoints.utils import get_mious
    from .AEF.ambiguity import ambiguity_function
    ambiguity_soft, ambiguity_count = ambiguity_function(p, posmask_test, nsample_test, neighbor_idx_test, cctype,
                                                         ccbeta, vis, nu)
    label, pred = label.reshape(-1), pred.reshape(-1)
    mapping = torch.floor(ambiguity_soft * 10 + 1)
    nu_m = nu * 10 + 1
    groups = [mapping == 1, (1 < mapping) & (mapping < nu_m), mapping == nu_m, (nu_m < mapping) & (mapping < 11),
              mapping == 11]
    mious, maccs, oas, counts = [], [], [], []
    for cm, g in zip((cm_0, cm_low, cm_semi, cm_high, cm_1), groups):
        cm.update(pred[g], label[g])
        miou, macc, oa, _, _ = get_mious(cm.tp, cm.union, cm.count)
        mious.append(round(miou, 2)); maccs.append(round(macc, 2)); oas.append(round(oa, 2))
        counts.append(cm.count.tolist())
    # accuracy per bin: one bincount over bin * correct (bin 0 collects the wrong predictions)
    bins = mapping.long()
    total = torch.bincount(bins, minlength=12)
    right = torch.bincount(bins * (pred == label).long(), minlength=12)
    total_l, right_l = total.tolist(), right.tolist()
    ratio = {float(k): right_l[k] / total_l[k] for k in range(1, 12) if total_l[k] > 0}
    cls = {}
    classes = torch.unique(label)
    lab_bins = torch.bincount(label * 12 + bins, minlength=(int(classes.max()) + 1) * 12).view(-1, 12).tolist()
    for c in classes.tolist():
        row = lab_bins[c]
        n = sum(row)
        shares = [row[1], sum(row[2:6]), row[6], sum(row[7:11]), row[11]]
        cls[c] = [round(s / n * 100, 2) for s in shares]
    return (ambiguity_soft, ratio, ambiguity_count, [1.0, 1.0, 1.0, 1.0, 1.0], cls, mious, maccs, oas, counts)


def ambiguity_summary(num_classes, ambiguity_vs_accuracy_list, ambiguity_vs_count_list, ambiguity_vs_accuracy_lowsemihigh_list,
                      ambiguity_vs_cls_list, ambiguity_cm_miou, ambiguity_cm_macc, ambiguity_cm_oa, ambiguity_cm_count):
    """Averages over the test clouds of what ambiguity_metrics returned for each (AMContrast3D/metrics.py:10-30; called
    once at the end of test_boundary_inner, examples/segmentation/main_AA.py:790-794).  Prints the tables like the
    reference and returns them: per-class shares per ambiguity group, group shares, and mIoU / mAcc / OA / per-class
    counts of the five groups (a = 0, low, semi, high, a = 1)."""
    import numpy as np
    per_class = {}
    for c in range(num_classes):
        rows = [per_cloud[c] for per_cloud in ambiguity_vs_cls_list if c in per_cloud]
        per_class[c] = np.around(np.mean(rows, axis=0), decimals=3) if rows else None
        print('count per cls: ', c, per_class[c])
    out = {
        'cls': per_class,
        'count': np.around(np.mean(ambiguity_vs_count_list, axis=0), decimals=3),
        'acc_low_semi_high': np.around(np.mean(ambiguity_vs_accuracy_lowsemihigh_list, axis=0), decimals=3),
        'miou': np.around(np.mean(ambiguity_cm_miou, axis=0), decimals=2),
        'macc': np.around(np.mean(ambiguity_cm_macc, axis=0), decimals=2),
        'oa': np.around(np.mean(ambiguity_cm_oa, axis=0), decimals=2),
        'count_per_class': np.around(np.mean(ambiguity_cm_count, axis=0), decimals=0),
    }
    print('count per a_i: 0, low=(0,0.5), semi=0.5, high=(0.5,1), 1:', out['count'])
    print('acc per a_i: 0, low=(0,0.5), semi=0.5, high=(0.5,1), 1:', out['acc_low_semi_high'])
    for key in ('miou', 'macc', 'oa'):
        print(f'{key} per ambiguity:', out[key])
    for name, row in zip(('0', 'low', 'semi', 'high', '1'), out['count_per_class']):
        print(f'count-{name}:', row)
    return out


def vis_tsne(stageACE_list_all):
    """t-SNE scatter plots of the stage embeddings (AMContrast3D/metrics.py:187-...): a visualisation aid that needs
    seaborn / matplotlib; not part of this package.  With AMC3D_REFERENCE_ROOT set the reference's own function runs."""
    import importlib.util
    import os
    root = os.environ.get('AMC3D_REFERENCE_ROOT')
    path = os.path.join(root, 'openpoints', 'AMContrast3D', 'metrics.py') if root else None
    if not path or not os.path.exists(path):
        raise NotImplementedError('vis_tsne is a plotting helper of the reference; set AMC3D_REFERENCE_ROOT to use it')
    spec = importlib.util.spec_from_file_location('_reference_amcontrast3d_metrics', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.vis_tsne(stageACE_list_all)
