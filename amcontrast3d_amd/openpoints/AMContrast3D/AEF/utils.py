"""Stage bookkeeping helpers (AMContrast3D/AEF/utils.py:11-52, 106-115)."""
import torch
import torch.nn.functional as F

from openpoints.cpp.pointops.functions import pointops


def get_subscene_label_CBL(stage_n, stage_i, stage_list, target, nstride, num_classes, ignore_index):
    """Per-point class distribution at stage ``stage_i``: one-hot labels at full resolution,
    and for a sub-sampled stage the mean one-hot label of the prod(nstride[:i]) nearest
    full-resolution points (utils.py:11-43).  With an ``ignore_index`` the ignored
    points form an extra class (:15-19)."""
    if ignore_index is not None:
        num_classes = num_classes + 1
        if (target == ignore_index).sum() > 0:
            target = target.clone()
            target[target == ignore_index] = num_classes - 1
    x = F.one_hot(target, num_classes)
    if stage_i == 0:
        return x.float()
    kr = int(torch.prod(nstride[:stage_i]))  # 4, 16, 64 for stride-4 stages
    src = stage_list['up'][0]
    dst = stage_list[stage_n][stage_i]
    neighbor_idx, _ = pointops.knnquery(kr, src['p_out'], dst['p_out'], src['offset'], dst['offset'])
    neighbor_idx = neighbor_idx.view(-1).long()
    x = x[neighbor_idx, :].view(dst['p_out'].shape[0], kr, x.shape[1])
    return x.float().mean(-2)


def get_subscene_class(stage_n, stage_i, stage_list, target, nstride, num_classes, ignore_index):
    """arg-max of get_subscene_label_CBL as int32 class ids (m), without materialising the
    (m,kr,ncls) one-hot gather: the majority vote runs in one kernel (lowest class id wins ties,
    as torch.argmax of the mean one-hot does)."""
    from amcontrast3d_amd import ops
    if ignore_index is not None:
        num_classes = num_classes + 1
        target = torch.where(target == ignore_index, torch.full_like(target, num_classes - 1), target)
    labels0 = target.to(torch.int32)
    if stage_i == 0:
        return labels0, num_classes
    kr = int(torch.prod(nstride[:stage_i]))
    src = stage_list['up'][0]
    dst = stage_list[stage_n][stage_i]
    knn = ops.knnquery_squared if labels0.is_cuda else pointops.knnquery  # (the distances are not used: no root taken)
    neighbor_idx, _ = knn(kr, src['p_out'], dst['p_out'], src['offset'], dst['offset'])
    return ops.vote_labels(labels0, neighbor_idx, num_classes), num_classes


def fetch_pxo(stage_n, stage_i, stage_list, ftype):
    stage = stage_list[stage_n][stage_i]
    return stage['p_out'], stage['f_out'], stage['offset']


def get_ftype(ftype):
    if ftype in ['out', 'fout', 'f_out', 'latent', 'logits', 'probs']:
        ptype = 'p_out'
        ftype = 'f_out' if ftype in ['out', 'fout'] else ftype
    elif ftype in ['sample', 'fsample', 'f_sample']:
        ptype = 'p_sample'
        ftype = 'f_sample' if ftype in ['sample', 'fsample'] else ftype
    else:
        raise KeyError(f'not supported ftype = {ftype}')
    return ftype, ptype
