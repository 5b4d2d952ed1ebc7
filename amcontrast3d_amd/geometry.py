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
def precompute_sampling(model, data, aux_stream=None, join=True):
    """Encoder / decoder geometry of the batch dict `data` (needs only 'pos'):
    -> {'encoder': per stage a list with one plan per block, 'decoder': per level 3-NN idx / weights}.

    The four FPS levels form the only serial chain (level i+1 samples level i's output); they run
    back to back on the current stream.  Ball queries, relative positions and 3-NN hang off that chain
    and, when an `aux_stream` is given, run there, concurrently with the later FPS levels.

    Under hipGraph capture fork `aux_stream` from the capture's origin stream yourself, like the
    calling stream, and pass join=False (join it at the origin): ROCm 7.2 segfaults while capturing a
    fork made from an already forked stream or a wait in both directions between two forks
    (scratch/graph_patterns.py)."""
    from openpoints.models.backbone.pointnext_AA import _segment_offset  # noqa: F401
    m = _unwrap(model)
    pos = data["pos"]
    cur = torch.cuda.current_stream(pos.device)
    aux = aux_stream if aux_stream is not None else cur
    stages_mod = list(m.encoder.encoder)
    p, enc = [pos], []
    for i, stage in enumerate(stages_mod):
        g = stage[0].plan_sample(p[-1])
        enc.append([g])
        p.append(g["new_p"])
        if aux is not cur:
            aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            stage[0].plan_group(p[i], g)
            cache = {}
            for blk in list(stage)[1:]:  # blocks of one stage share one self-query per (radius, nsample)
                grouper = blk.convs.grouper
                key = (getattr(grouper, "radius", None), getattr(grouper, "nsample", None))
                if key not in cache:
                    cache[key] = blk.plan(p[i + 1])
                enc[i].append(cache[key])
    with torch.cuda.stream(aux):
        dec = m.decoder.plan_geometry(p)
    if aux is not cur and join:
        cur.wait_stream(aux)
    return {"encoder": enc, "decoder": dec}


@torch.no_grad()
def precompute_fps(model, data):
    """The serial part only: per stage {'fps_idx', 'new_p'} (four dependent FPS launches + gathers)."""
    m = _unwrap(model)
    p, out = data["pos"], []
    for stage in m.encoder.encoder:
        g = stage[0].plan_sample(p)
        out.append(g)
        p = g["new_p"]
    return out


@torch.no_grad()
def precompute_fps_levels(model, p, first, last):
    """Levels [first, last) of the sampling chain starting from the cloud `p` that level `first` samples
    (data['pos'] for first == 0, the previous level's 'new_p' otherwise)."""
    m = _unwrap(model)
    out = []
    for stage in list(m.encoder.encoder)[first:last]:
        g = stage[0].plan_sample(p)
        out.append(g)
        p = g["new_p"]
    return out


@torch.no_grad()
def precompute_rest(model, contrast_head, data, fps, num_classes, ignore_index, ambiguity_args):
    """Everything that hangs off a finished sampling `fps` (from precompute_fps): ball queries, relative
    positions, 3-NN, loss geometry.  -> a full plan whose per-stage dicts also hold fps's tensors."""
    m = _unwrap(model)
    p, enc = [data["pos"]], []
    for i, stage in enumerate(m.encoder.encoder):
        g = dict(fps[i])
        stage[0].plan_group(p[i], g)
        blocks, cache = [g], {}
        p.append(g["new_p"])
        for blk in list(stage)[1:]:
            grouper = blk.convs.grouper
            key = (getattr(grouper, "radius", None), getattr(grouper, "nsample", None))
            if key not in cache:
                cache[key] = blk.plan(p[i + 1])
            blocks.append(cache[key])
        enc.append(blocks)
    plan = {"encoder": enc, "decoder": m.decoder.plan_geometry(p)}
    from . import ops
    with ops.knn_grid_reuse():  # the refinement's k-NN searches the same clouds as the loss: shared cell grids
        plan["loss"] = precompute_loss(contrast_head, plan, data["y"], num_classes, ignore_index, ambiguity_args)
        refine = precompute_refine(m, plan)
        if refine is not None:
            plan["refine"] = refine
    return plan


@torch.no_grad()
def precompute_refine(model, plan):
    """AMContrast3D++ only: the neighbour lists of the decoder's masked refinement (MaskedRefine.py:60-70, k-NN of
    every decoder level's cloud over the whole batch, self match dropped) -- coordinates only, so part of the plan.
    -> {level i in -1..-4: idx (B*n, K-1) int32} or None for models without refinement."""
    from . import ops
    from openpoints.models.backbone.pointnext_AA import _segment_offset
    k = getattr(model, "nsample_k", None)
    if k is None or getattr(model, "linear_mapping", False):
        return None
    out = {}
    stages = stage_points(plan)  # flattened clouds of p[1..4]; decoder level i refines p[i-1] = stage index 4+i
    for i in range(-1, -len(stages) - 1, -1):
        xyz = stages[4 + i]["p_out"]
        o = _segment_offset(xyz.shape[0], xyz.device)
        idx, _ = ops.knnquery_squared(k, xyz, xyz, o, o)  # (the distances are not used: no root taken)
        out[i] = idx[..., 1:].contiguous()
    return out


_FPS_KEYS = ("fps_idx", "new_p", "fps_idx32")  # what the sampling chain produces (the serial part of a plan)


def split(plan):
    """(fps part, rest part) of a full plan, as structures referencing the plan's own tensors."""
    fps = [{k: b[0][k] for k in _FPS_KEYS if k in b[0]} for b in plan["encoder"]]
    rest = {"encoder": [[{k: v for k, v in b[0].items() if k not in _FPS_KEYS}] + list(b[1:])
                        for b in plan["encoder"]],
            "decoder": plan["decoder"], "loss": plan["loss"]}
    if "refine" in plan:
        rest["refine"] = plan["refine"]
    return fps, rest


def join(fps, rest):
    """Inverse of split(): a full plan whose dicts reference the tensors of `fps` and `rest` (no copies)."""
    plan = {"encoder": [[dict(f, **b[0])] + list(b[1:]) for f, b in zip(fps, rest["encoder"])],
            "decoder": rest["decoder"], "loss": rest["loss"]}
    if "refine" in rest:
        plan["refine"] = rest["refine"]
    return plan


def stage_points(plan):
    """The flattened clouds of the loss stages of a sampling plan: [{'p_out' (B*n,3), 'offset'}] x 4
    (what pointnext_AA.py:458-462 puts into stageACE_list)."""
    from openpoints.models.backbone.pointnext_AA import _segment_offset
    out = []
    for blocks in plan["encoder"][:-1]:
        flat = torch.flatten(blocks[0]["new_p"], start_dim=0, end_dim=1)
        out.append({"p_out": flat, "offset": _segment_offset(flat.shape[0], flat.device)})
    return out


@torch.no_grad()
def precompute_loss(contrast_head, plan, y, num_classes, ignore_index, ambiguity_args):
    """Loss geometry (per stage: k-NN, class vote, positive mask, ambiguity) of the batch whose
    sampling plan is `plan`; needs only coordinates and labels."""
    stages = stage_points(plan)
    return contrast_head.plan(y, {"up": stages, "down": stages}, num_classes, ignore_index, ambiguity_args)


@torch.no_grad()
def precompute(model, contrast_head, data, num_classes, ignore_index, ambiguity_args, aux_stream=None, join=True):
    """-> {'encoder', 'decoder', 'loss'}: the whole coordinate-only half of a step for `data` (pos, y)."""
    plan = precompute_sampling(model, data, aux_stream=aux_stream, join=True if aux_stream is None else join)
    if aux_stream is not None and not join:
        with torch.cuda.stream(aux_stream):
            plan["loss"] = precompute_loss(contrast_head, plan, data["y"], num_classes, ignore_index, ambiguity_args)
    else:
        plan["loss"] = precompute_loss(contrast_head, plan, data["y"], num_classes, ignore_index, ambiguity_args)
    return plan


def clone(plan):
    """Deep copy of a plan's tensors (fresh static buffers); views are cloned as contiguous tensors
    except the loss stages' neighbour index, which keeps its idx[:, 1:] view layout."""
    if torch.is_tensor(plan):
        if plan.dim() == 2 and plan.stride(1) == 1 and plan.stride(0) == plan.shape[1] + 1 and plan.storage_offset() == 1:
            full = torch.empty(plan.shape[0], plan.shape[1] + 1, dtype=plan.dtype, device=plan.device)
            full[:, 1:].copy_(plan)
            return full[:, 1:]
        return plan.clone()
    if isinstance(plan, dict):
        return {k: clone(v) for k, v in plan.items()}
    if isinstance(plan, (list, tuple)):
        return type(plan)(clone(v) for v in plan)
    return plan


def _walk(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _walk(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_walk(v, fn) for v in obj)
    return obj


def _pairs(dst, src, out):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():
            out.append((dst, src))
    elif isinstance(dst, dict):
        for k in dst:
            _pairs(dst[k], src[k], out)
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _pairs(d, s, out)


def copy_into(dst, src):
    """In-place copy of every tensor of plan `src` into the same-shaped plan `dst` (static buffers
    for graph replay).  Views (e.g. idx[:, 1:]) are copied through their storage like any tensor.
    The copies of one call go out as multi-tensor launches (one per dtype), not one launch per tensor: a plan holds
    ~40 tensors and the pipeline's rotate step copies four plans between two steps of the training stream."""
    import os
    pairs = []
    _pairs(dst, src, pairs)
    if os.environ.get("AMC3D_NO_FOREACH_COPY"):
        for d, s in pairs:
            d.copy_(s)
        return
    by_dtype = {}
    for d, s in pairs:
        if d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and d.shape == s.shape:
            by_dtype.setdefault(d.dtype, ([], []))
            by_dtype[d.dtype][0].append(d)
            by_dtype[d.dtype][1].append(s)
        else:
            d.copy_(s)
    for ds, ss in by_dtype.values():
        torch._foreach_copy_(ds, ss)
