"""ORACLE -- test infrastructure only (imported by tests/ and oracle/gen_golden.py, never by the product).

CPU restatement of the reference's evaluation arithmetic around the hot path:

    confusion / metrics      openpoints/utils/metrics.py:50-181   (ConfusionMatrix.update, all_metrics, get_mious)
    posmask_searching        openpoints/AMContrast3D/metrics.py:160-184
    boundary test            examples/segmentation/main_AA.py:470-476, 628-633
    voxel partition          examples/segmentation/main_AA.py:91-116 + openpoints/dataset/data_util.py:92-143
    scatter mean             torch_scatter.scatter(reduce='mean') at main_AA.py:662: torch_scatter 2.x is not installed
                             here; its documented mean is sum / max(count, 1) per index
    whole-cloud test loop    examples/segmentation/main_AA.py:556-684 (one cloud)

Pinned by tests/golden/eval_w8_room.npz, recorded from the reference's own model (eval mode), ConfusionMatrix,
get_mious, posmask_searching and voxelize (oracle/gen_golden.py eval).  examples/segmentation/main_AA.py itself cannot
be imported in the build container (wandb / torch_scatter / tensorboard missing), so its loop body is restated from
the text and only its ingredients are pinned: "loop parity unpinned beyond its ingredients".
Plain numpy, integer work bit-exact.
"""
import numpy as np
import torch

from . import model_ref
from . import pointops_ref as K


def confusion(pred, true, num_classes, ignore_index=None):
    """(num_classes, num_classes) int64: rows = true class, columns = predicted (metrics.py:62-73)."""
    pred, true = np.asarray(pred).reshape(-1).astype(np.int64), np.asarray(true).reshape(-1).astype(np.int64)
    v = num_classes + (1 if ignore_index is not None else 0)
    if ignore_index is not None:
        ign = true == ignore_index
        pred, true = np.where(ign, v - 1, pred), np.where(ign, v - 1, true)
    m = np.zeros((v, v), dtype=np.int64)
    np.add.at(m, (true, pred), 1)
    return m[:num_classes, :num_classes]


def tp_union_count(m):
    tp = np.diag(m)
    return tp, m.sum(0) + m.sum(1) - tp, m.sum(1)


def get_mious(tp, union, count):
    """metrics.py:173-181, float32 like the torch original (int64 + 1e-10 promotes to float32 there)."""
    tp32, un32, ct32 = (np.asarray(a, dtype=np.float32) for a in (tp, union, count))
    eps = np.float32(1e-10)
    iou = (tp32 + eps) / (un32 + eps) * np.float32(100)
    acc = (tp32 + eps) / (ct32 + eps) * np.float32(100)
    oa = np.float32(np.asarray(tp).sum()) / np.float32(np.asarray(count).sum()) * np.float32(100)
    return float(iou.mean(dtype=np.float32)), float(acc.mean(dtype=np.float32)), float(oa), iou, acc


def all_metrics(m):
    """ConfusionMatrix.all_metrics (metrics.py:157-170): clamp(min=1) denominators, per cent."""
    tp, union, count = tp_union_count(m)
    iou = (tp / np.maximum(union, 1)).astype(np.float32) * np.float32(100)
    acc = (tp / np.maximum(count, 1)).astype(np.float32) * np.float32(100)
    oa = np.float32(tp.sum() / m.sum()) * np.float32(100)
    return float(iou.mean(dtype=np.float32)), float(acc.mean(dtype=np.float32)), float(oa), iou, acc


def posmask_searching(xyz, target, nsample, num_classes, ignore_index):
    """-> posmask (m, nsample-1) bool, neighbor_idx (m, nsample-1) int32 (AMContrast3D/metrics.py:160-184)"""
    xyz = torch.as_tensor(xyz, dtype=torch.float32).contiguous()
    target = np.asarray(target).reshape(-1).astype(np.int64)
    if ignore_index is not None:
        target = np.where(target == ignore_index, num_classes, target)
    o = torch.tensor([xyz.shape[0]], dtype=torch.int32)
    idx, _ = K.knnquery(nsample, xyz, xyz, o, o)
    idx = idx[:, 1:].numpy()
    return target[:, None] == target[idx], idx


def boundary_mask(xyz, target, nsample, num_classes, ignore_index):
    same = posmask_searching(xyz, target, nsample, num_classes, ignore_index)[0].sum(-1)
    return (0 < same) & (same < nsample)


def voxel_parts(coord, voxel_size, shuffle_rng=None):
    """multi_voxel sub-clouds (main_AA.py:95-113): FNV-1a hash of the voxel coordinates, stable order by hash, pass i
    takes the (i mod count)-th point of each voxel."""
    cells = np.floor(np.asarray(coord) / voxel_size).astype(np.uint64)
    h = np.uint64(14695981039346656037) * np.ones(len(cells), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(3):
            h = (h * np.uint64(1099511628211)) ^ cells[:, j]
    order = np.argsort(h)
    _, count = np.unique(h[order], return_counts=True)
    first = np.concatenate([[0], np.cumsum(count)[:-1]])
    parts = []
    for i in range(count.max()):
        part = order[first + i % count]
        if shuffle_rng is not None:
            shuffle_rng.shuffle(part)
        parts.append(part)
    return parts


def scatter_mean(src, index, size):
    out = np.zeros((size, src.shape[1]), dtype=np.float32)
    np.add.at(out, index, src)
    cnt = np.zeros(size, dtype=np.float32)
    np.add.at(cnt, index, np.float32(1))
    return out / np.maximum(cnt, 1)[:, None]


def test_cloud(sd, cfg, coord, feat, label, parts, num_classes, ignore_index, nsample):
    """One cloud of test_boundary_inner (main_AA.py:556-684): per-part logits (eval-mode model), voted logits,
    prediction and the three confusion matrices."""
    logits_parts, pb, pi, tb, ti = [], [], [], [], []
    with torch.no_grad():
        for part in parts:
            cp = coord[part] - coord[part].min(0)
            pos = torch.from_numpy(np.ascontiguousarray(cp, dtype=np.float32)).unsqueeze(0)
            x = torch.from_numpy(np.ascontiguousarray(np.concatenate([feat[part], cp[:, 2:3]], 1).T, dtype=np.float32)).unsqueeze(0)
            lg = model_ref.model_forward(sd, cfg, {"pos": pos, "x": x}, training=False)[0][0].numpy()  # (ncls, n)
            logits_parts.append(lg)
            b = boundary_mask(pos[0], label[part], nsample, num_classes, ignore_index)
            pr = lg.argmax(0)
            pb.append(pr[b]); pi.append(pr[~b]); tb.append(label[part][b]); ti.append(label[part][~b])
    flat = np.concatenate([lg.T for lg in logits_parts], 0)
    voted = scatter_mean(flat, np.hstack(parts), len(label))
    pred = voted.argmax(1)
    return {"logits_parts": logits_parts, "voted": voted, "pred": pred,
            "cm": confusion(pred, label, num_classes, ignore_index),
            "cm_b": confusion(np.concatenate(pb), np.concatenate(tb), num_classes, ignore_index),
            "cm_i": confusion(np.concatenate(pi), np.concatenate(ti), num_classes, ignore_index)}


def ambiguity_metrics(p, label, pred, nsample, num_classes, ignore_index, beta, nu):
    """AMContrast3D/metrics.py:33-157 for cctype Method2: bins, five confusion matrices, accuracy per bin and the
    per-class shares; the a_i come from model_ref.ambiguity (AEF/ambiguity.py:11-71)."""
    posmask, nidx = posmask_searching(p, label, nsample, num_classes, ignore_index)
    a = model_ref.ambiguity(torch.as_tensor(p, dtype=torch.float32), torch.from_numpy(posmask), torch.from_numpy(nidx),
                            beta).numpy()
    mapping = np.floor(a * np.float32(10) + np.float32(1)).astype(np.int64)
    nu_m = nu * 10 + 1
    groups = [mapping == 1, (1 < mapping) & (mapping < nu_m), mapping == nu_m, (nu_m < mapping) & (mapping < 11), mapping == 11]
    mats = [confusion(pred[g], label[g], num_classes, ignore_index) for g in groups]
    ratio = {}
    for k in np.unique(mapping):
        sel = mapping == k
        ratio[float(k)] = float((pred[sel] == label[sel]).sum()) / float(sel.sum())
    cls = {}
    for c in np.unique(label):
        mc = mapping[label == c]
        n = len(mc)
        cls[int(c)] = [round(float(s) / n * 100, 2) for s in
                       ((mc == 1).sum(), ((1 < mc) & (mc < 6)).sum(), (mc == 6).sum(), ((6 < mc) & (mc < 11)).sum(), (mc == 11).sum())]
    return {"a": a, "mapping": mapping, "mats": mats, "ratio": ratio, "cls": cls}
