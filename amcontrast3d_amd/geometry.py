"""Coordinate-only half of a training step, computable ahead of the feature half.

FPS picks, ball-query indices and relative positions, 3-NN indices / weights, the loss stages'
k-NN, positive masks and ambiguities depend only on ``pos`` and ``y`` -- not on features or weights.
``precompute`` builds all of it for one batch; ``model(data)`` and the criterion consume it when the
batch dict carries it under ``'_geometry'`` and build it themselves otherwise (one code path).

Why: the FPS chain (4 dependent launches, one workgroup per cloud) keeps 8 of the 256 CUs busy for
most of the geometry time.  Preparing batch k+1's geometry on a side stream while batch k runs its
convolutions hides it completely (bench.py does this inside the captured graph).
"""
import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


@torch.no_grad()
def precompute(model, contrast_head, data, num_classes, ignore_index, ambiguity_args):
    """-> {'encoder': ..., 'decoder': ..., 'loss': ...} for the batch dict `data` (pos, y)."""
    m = _unwrap(model)
    pos = data["pos"]
    enc = m.encoder.plan_geometry(pos)
    p = [pos] + [blocks[0]["new_p"] for blocks in enc]
    dec = m.decoder.plan_geometry(p)
    from openpoints.models.backbone.pointnext_AA import _segment_offset
    stages = []
    for q in p[1:-1]:
        flat = torch.flatten(q, start_dim=0, end_dim=1)
        stages.append({"p_out": flat, "offset": _segment_offset(flat.shape[0], flat.device)})
    loss = contrast_head.plan(data["y"], {"up": stages, "down": stages}, num_classes, ignore_index, ambiguity_args)
    return {"encoder": enc, "decoder": dec, "loss": loss}


def _walk(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _walk(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_walk(v, fn) for v in obj)
    return obj


def copy_into(dst, src):
    """In-place copy of every tensor of plan `src` into the same-shaped plan `dst` (static buffers
    for graph replay).  Views (e.g. idx[:, 1:]) are copied through their storage like any tensor."""
    if torch.is_tensor(dst):
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            copy_into(d, s)
