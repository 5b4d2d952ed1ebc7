"""Evaluation loops around the hot path: validation batches and whole-room testing with boundary / inner mIoU
(the reference's `validate_boundary_inner` and `test_boundary_inner`, examples/segmentation/main_AA.py:431-802;
SURVEY.md section 8(f) rank 2).

The reference's loops are welded to its dataset readers, transform registry, tqdm / wandb logging and result
files; what they compute is restated here on arrays the caller already holds:

    voxel_parts                 main_AA.py:91-116 (`load_data`, multi_voxel mode) + dataset/data_util.py:92-143
    voxel_representatives       main_AA.py:99-107, 666 (`load_data`, nearest_neighbor mode)
    boundary_mask               main_AA.py:470-476, 628-633 (posmask_searching + the 0 < n+ < nsample test)
    scatter_mean                torch_scatter.scatter(..., reduce='mean') of main_AA.py:662 (package absent here)
    validate_boundary_inner     main_AA.py:431-513
    test_cloud_boundary_inner   one iteration of the cloud loop of main_AA.py:556-684
    summarize                   main_AA.py:484-506 / 746-770 (get_mious over the accumulated matrices)

The model runs with model.eval(): BatchNorm uses its running statistics on the fused inference kernels
(ops.bn_eval), FPS / ball query / grouped convolutions / 3-NN are the same HIP kernels as in training.
"""
import numpy as np
import torch

from . import activate


def _fnv64(cells):
    """FNV-1a over the integer voxel coordinates, as dataset/data_util.py:92-105 (uint64 wrap-around)."""
    h = np.full(cells.shape[0], 14695981039346656037, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(cells.shape[1]):
            h *= np.uint64(1099511628211)
            h ^= cells[:, j]
    return h


def voxel_parts(coord, voxel_size, rng=None):
    """Split a cloud into sub-clouds holding one point per occupied voxel each (test_mode 'multi_voxel'):
    part i takes the (i mod count)-th point of every voxel, so count.max() parts cover every point and sparse voxels
    repeat theirs.  Returns the list of index arrays (each shuffled with `rng`, as the reference shuffles them)."""
    cells = np.floor(np.asarray(coord) / np.asarray(voxel_size)).astype(np.uint64)
    key = _fnv64(cells)
    order = np.argsort(key)
    _, count = np.unique(key[order], return_counts=True)
    first = np.cumsum(np.insert(count, 0, 0)[:-1])
    rng = np.random.default_rng(0) if rng is None else rng
    parts = []
    for i in range(int(count.max())):
        part = order[first + i % count]
        rng.shuffle(part)
        parts.append(part)
    return parts


def voxel_representatives(coord, voxel_size, rng=None):
    """test_mode 'nearest_neighbor' (main_AA.py:99-107): ONE sub-cloud holding a random point of every occupied voxel;
    afterwards every point takes the logits of its voxel's representative (main_AA.py:666).
    Returns (part, expand): the representatives' indices (shuffled) and, for every point of the cloud, the position in
    `part` of its voxel's representative -- test_cloud_boundary_inner(..., parts=[part], expand=expand)."""
    cells = np.floor(np.asarray(coord) / np.asarray(voxel_size)).astype(np.uint64)
    key = _fnv64(cells)
    order = np.argsort(key)
    _, voxel_of_sorted, count = np.unique(key[order], return_inverse=True, return_counts=True)
    first = np.cumsum(np.insert(count, 0, 0)[:-1])
    rng = np.random.default_rng(0) if rng is None else rng
    pick = order[first + rng.integers(0, count.max(), count.size) % count]  # one point per voxel, voxel order
    perm = rng.permutation(count.size)
    part = pick[perm]
    where = np.argsort(perm)                       # voxel -> position of its representative in `part`
    expand = np.empty(len(key), dtype=np.int64)
    expand[order] = where[voxel_of_sorted]         # undo the sort by hash
    return part, expand


@torch.no_grad()
def boundary_mask(xyz, target, nsample, num_classes, ignore_index):
    """(m) bool: points whose nsample-1 nearest neighbours contain between 1 and nsample-1 points of their own class.
    (The upper bound is `< nsample`, not `< nsample - 1`, exactly as main_AA.py:475 writes it: a point all of whose
    neighbours agree with it therefore counts as boundary too, and "inner" means no neighbour agrees.)"""
    activate()
    from openpoints.AMContrast3D.metrics import posmask_searching
    posmask, _ = posmask_searching(xyz, target, nsample, num_classes, ignore_index)
    same = posmask.sum(-1)
    return torch.logical_and(0 < same, same < nsample)


def scatter_mean(src, index, size=None):
    """out[j] = mean of src[i] over index[i] == j (0 where no i): torch_scatter.scatter(src, index, dim=0,
    reduce='mean') for 2-d src, which is how overlapping sub-cloud logits are voted (main_AA.py:662)."""
    n = int(index.max().item()) + 1 if size is None else size
    out = torch.zeros(n, src.shape[1], dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


def _matrices(num_classes, ignore_index):
    activate()
    from openpoints.utils import ConfusionMatrix
    return [ConfusionMatrix(num_classes=num_classes, ignore_index=ignore_index) for _ in range(3)]


def summarize(cm, cm_b=None, cm_i=None, distributed=False):
    """(miou, macc, oa, ious, accs) for the whole set [+ the same five for boundary and inner points], with the
    cross-rank sum of (tp, union, count) first when `distributed` (main_AA.py:459-462, 484-500)."""
    activate()
    from openpoints.utils import get_mious
    out = ()
    for m in (cm, cm_b, cm_i):
        if m is None:
            continue
        tp, union, count = m.tp, m.union, m.count
        if distributed:
            import torch.distributed as dist
            from .graphs import on_side_stream
            on_side_stream(lambda: (dist.all_reduce(tp), dist.all_reduce(union), dist.all_reduce(count)))
        out += tuple(get_mious(tp, union, count))
    return out


@torch.no_grad()
def validate_boundary_inner(model, batches, num_classes, ignore_index, nsample, miou_B_I=True, distributed=False):
    """Validation pass over batches of ONE cloud each (the reference's val loader uses batch size 1;
    `data['pos'].squeeze()` at main_AA.py:469 assumes it): dicts with pos (1,N,3), x (1,C,N), y (1,N) on the GPU.
    Returns summarize(...) of the whole / boundary / inner confusion matrices."""
    model.eval()
    cm, cm_b, cm_i = _matrices(num_classes, ignore_index)
    for data in batches:
        target = data["y"].reshape(data["y"].shape[0], -1)
        logits, _ = model(data)
        pred = logits.argmax(dim=1)
        cm.update(pred, target)
        if miou_B_I:
            assert data["pos"].shape[0] == 1, "boundary / inner split is defined per cloud (batch size 1)"
            b = boundary_mask(data["pos"][0], target[0], nsample, num_classes, ignore_index).unsqueeze(0)
            cm_b.update(pred[b], target[b])
            cm_i.update(pred[~b], target[~b])
    return summarize(cm, cm_b, cm_i, distributed) if miou_B_I else summarize(cm, distributed=distributed)


@torch.no_grad()
def boundary_masks_stacked(pos, labels, nsample, num_classes, ignore_index):
    """boundary_mask for P equally sized clouds in one k-NN call: pos (P,n,3), labels (P,n) -> (P,n) bool.  The clouds
    are the segments of one knnquery (offsets n, 2n, ...), which searches every segment on its own -- the same lists
    as P separate calls (main_AA.py:628 calls posmask_searching per sub-cloud)."""
    activate()
    from openpoints.cpp.pointops.functions import pointops
    from . import ops
    P, n = labels.shape
    xyz = pos.reshape(P * n, 3).contiguous()
    target = labels.reshape(-1)
    if ignore_index is not None:
        target = torch.where(target == ignore_index, num_classes, target)
    o = torch.arange(1, P + 1, dtype=torch.int32, device=xyz.device) * n
    idx, _ = pointops.knnquery(nsample, xyz, xyz, o, o)
    same = ops.posmask_from_labels(target.to(torch.int32).contiguous(), idx[..., 1:].contiguous()).sum(-1)
    return torch.logical_and(0 < same, same < nsample).view(P, n)


@torch.no_grad()
def test_cloud_boundary_inner(model, coord, feat, label, parts, num_classes, ignore_index, nsample,
                              make_input=None, miou_B_I=True, batch=64, expand=None):
    """One whole cloud (room): every sub-cloud of `parts` goes through the model, overlapping logits are averaged per
    point, and three confusion matrices are filled -- all points (voted prediction), boundary and inner points (per
    sub-cloud predictions, as the reference keeps them: main_AA.py:634-641, 651-657, 671-676).

    coord (n,3) float array, feat (n,c) float array or None, label (n) int tensor on the GPU, parts: index arrays.
    make_input(coord_part, feat_part) -> model input dict; default: pos shifted to its minimum corner, x = feat and
    the height channel (the S3DIS feature_keys 'x,heights', cfgs/s3dis/default.yaml).
    batch: sub-clouds stacked per model call.  The reference feeds them one at a time (main_AA.py:575-611); in eval
    mode nothing couples the clouds of a batch (BatchNorm uses running statistics, FPS / ball query / 3-NN work per
    cloud), so the logits are the same -- but the FPS chain is latency-bound, 8 ms for 24 k points on ONE workgroup
    per cloud, and 5/6 of the model time at 8 clouds per call (scratch/eval_phases.py): stacking a whole room runs all
    its chains side by side.  Sub-clouds of a voxel partition have one point per voxel, hence equal size; with the
    default make_input they are also gathered and shifted on the GPU (one upload of the room instead of one per
    sub-cloud) and their boundary masks come from one segmented k-NN call.  Ragged parts fall back to single calls.
    expand: with the single sub-cloud of voxel_representatives, the per-point index into it ('nearest_neighbor' mode).
    Returns dict(pred, logits, cm, cm_b, cm_i)."""
    model.eval()
    dev = label.device
    cm, cm_b, cm_i = _matrices(num_classes, ignore_index)
    same_size = len({len(p) for p in parts}) == 1
    step = max(1, int(batch)) if same_size else 1
    index = torch.from_numpy(np.hstack(parts)).to(dev)
    all_logits, pb, pi, tb, ti = [], [], [], [], []

    def split_by_boundary(pred_stack, label_stack, b):
        pb.append(pred_stack[b]); pi.append(pred_stack[~b]); tb.append(label_stack[b]); ti.append(label_stack[~b])

    if make_input is None and same_size:
        # the room goes to the GPU once; sub-clouds are gathered, shifted to their minimum corner (in the array's own
        # precision, as numpy does it at main_AA.py:578-579) and stacked there
        idx = index.view(len(parts), -1)
        room = torch.from_numpy(np.ascontiguousarray(coord)).to(dev)
        room_f = None if feat is None else torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32)).to(dev)
        for j0 in range(0, len(parts), step):
            sel = idx[j0:j0 + step]
            pos = room[sel]
            pos = (pos - pos.amin(dim=1, keepdim=True)).float()
            cols = ([room_f[sel]] if room_f is not None else []) + [pos[..., 2:3]]
            data = {"pos": pos.contiguous(), "x": torch.cat(cols, dim=2).transpose(1, 2).contiguous()}
            logits, _ = model(data)
            all_logits.append(logits)
            if miou_B_I:
                split_by_boundary(logits.argmax(dim=1), label[sel],
                                  boundary_masks_stacked(data["pos"], label[sel], nsample, num_classes, ignore_index))
    else:
        if make_input is None:
            def make_input(coord_part, feat_part):
                pos = torch.from_numpy(np.ascontiguousarray(coord_part, dtype=np.float32)).to(dev).unsqueeze(0)
                cols = [torch.from_numpy(np.ascontiguousarray(feat_part, dtype=np.float32)).to(dev)] if feat_part is not None else []
                return {"pos": pos, "x": torch.cat(cols + [pos[0, :, 2:3]], dim=1).t().contiguous().unsqueeze(0)}
        inputs = []
        for part in parts:
            coord_part = np.asarray(coord)[part]
            inputs.append(make_input(coord_part - coord_part.min(0), None if feat is None else np.asarray(feat)[part]))
        for j0 in range(0, len(parts), step):
            chunk = inputs[j0:j0 + step]
            data = chunk[0] if len(chunk) == 1 else {k: torch.cat([d[k] for d in chunk], dim=0) for k in chunk[0]}
            logits, _ = model(data)
            all_logits.append(logits)
            if miou_B_I:
                for j, d in enumerate(chunk):
                    label_part = label[torch.from_numpy(parts[j0 + j]).to(dev)]
                    b = boundary_mask(d["pos"][0], label_part, nsample, num_classes, ignore_index)
                    split_by_boundary(logits[j].argmax(dim=0), label_part, b)
    flat = torch.cat([lg.transpose(1, 2).reshape(-1, num_classes) for lg in all_logits], dim=0)
    if expand is not None:
        assert len(parts) == 1, "expand belongs to the single sub-cloud of voxel_representatives"
        voted = flat[torch.from_numpy(np.asarray(expand)).to(dev)]
    else:
        voted = scatter_mean(flat, index, size=label.shape[0]) if len(parts) > 1 else flat[torch.argsort(index)]
    pred = voted.argmax(dim=1)
    cm.update(pred, label)
    if miou_B_I:
        cm_b.update(torch.cat(pb), torch.cat(tb))
        cm_i.update(torch.cat(pi), torch.cat(ti))
    return {"pred": pred, "logits": voted, "cm": cm, "cm_b": cm_b, "cm_i": cm_i}
