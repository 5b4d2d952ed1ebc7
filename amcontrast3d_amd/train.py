"""The training loop around the hot path: `train_one_epoch` of examples/segmentation/main_AA.py:370-428, the direct
caller of model + criterion, with the same arguments, the same per-iteration order of operations and the same return
value (loss average, mIoU, mAcc, OA, per-class IoU / accuracy of the training predictions).

What differs from the reference's loop, with identical arithmetic per batch:
  * the step is pipeline.GraphPipeline: hipGraph replays on three hardware queues -- the coordinate-only half of every step
    (FPS, ball queries, 3-NN, the loss's k-NN / votes / ambiguities) runs for the NEXT batches on two side queues while the
    current batch runs forward + loss + backward + clip + optimizer step as captured graphs on the training stream.  The
    graphs are built once per (model, optimizer, criterion, batch shape) from the first batch -- parameters, buffers and
    optimizer state are restored after the warm-up -- and reused by later epochs.  Loops the graphs cannot express run
    eagerly behind pipeline.GeometryPrefetcher, as before: gradient accumulation (step_per_update > 1), use_amp with a
    GradScaler, DistributedDataParallel-wrapped models, a batch whose shape differs from the first one's,
    cfg.graph_pipeline = False or AMC3D_EAGER_TRAIN=1;
  * the loss is accumulated on the device and read back once per epoch (the reference calls `loss.item()` every
    iteration, main_AA.py:419, which drains the GPU each step); `print_freq` progress lines therefore show the loss of
    the last *completed* read-back;
  * no tqdm / wandb.
Data: each batch is the reference's collated dict -- 'pos' (B,N,3), 'y' (B,N) or (B,N,1), and the keys named by
`cfg.feature_keys` ('x', 'heights', ...) point-major, as its datasets produce them (dataset/data_util.py:177-189).
"""
import itertools
import os

import torch

from . import activate
from .pipeline import GeometryPrefetcher, GraphPipeline

_PIPELINES = {}  # (model, optimizer, criterion, batch shape) -> (GraphPipeline, its training stream)


def release_pipelines():
    """drop the captured graphs (and the references to the models / optimizers they were built for)"""
    _PIPELINES.clear()


def get_features_by_keys(data, keys="pos,x"):
    """(B,N,c1), (B,N,c2), ... -> (B, c1+c2+..., N) contiguous: the model's 'x' (dataset/data_util.py:177-189)"""
    parts = [data[k] for k in keys.split(",")]
    x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)
    return x.transpose(1, 2).contiguous()


def _to_device_batches(train_loader, cfg, device):
    for data in train_loader:
        for key in list(data.keys()):
            if torch.is_tensor(data[key]):
                data[key] = data[key].to(device, non_blocking=True)
        data["y"] = data["y"].squeeze(-1) if data["y"].dim() == 3 else data["y"]
        data["x"] = get_features_by_keys(data, cfg.feature_keys)
        yield data


def train_one_epoch_mm(model, train_loader, criterion, optimizer, scheduler, scaler, epoch, cfg, prefetch_depth=2,
                       device=None):
    """The AMContrast3D++ loop (examples/segmentation/main_MM.py:370-449): the model returns (logits, stage list,
    refine rate), CrossEntropyAcePre returns (segmentation, w1 CE, w2 contrast, w3 regression) and the step minimises
    segmentation + regression.  Returns the reference's tuple: averages of (loss, segmentation, CE, contrast,
    regression, refine rate), then mIoU, mAcc, OA, per-class IoU / accuracy."""
    activate()
    from openpoints.AMContrast3D import MaskedRefine

    def step_loss(data, target):
        logits, stage, rate = model(data)
        seg, ce, am, reg = criterion(logits, target, stage, cfg.num_classes, cfg.ignore_index, cfg.ambiguity_args)
        rate = rate if torch.is_tensor(rate) else torch.tensor(float(rate), device=logits.device)
        return logits, seg + reg, (seg, ce, am, reg, rate)

    keep = MaskedRefine.RATE_ON_DEVICE
    MaskedRefine.RATE_ON_DEVICE = True  # no .item() per refinement stage: the loop stays ahead of the GPU
    try:
        return _run_epoch(model, train_loader, criterion, optimizer, scheduler, scaler, epoch, cfg, prefetch_depth, device,
                          step_loss, extras=5)
    finally:
        MaskedRefine.RATE_ON_DEVICE = keep


def train_one_epoch(model, train_loader, criterion, optimizer, scheduler, scaler, epoch, cfg, prefetch_depth=2,
                    device=None):
    """One pass over `train_loader`.  cfg needs num_classes, ignore_index, ambiguity_args, feature_keys, use_amp,
    step_per_update, grad_norm_clip (None / 0: off), sched_on_epoch -- the fields main_AA.py reads."""
    def step_loss(data, target):
        logits, stage = model(data)
        return logits, criterion(logits, target, stage, cfg.num_classes, cfg.ignore_index, cfg.ambiguity_args), ()

    return _run_epoch(model, train_loader, criterion, optimizer, scheduler, scaler, epoch, cfg, prefetch_depth, device,
                      step_loss, extras=0)


def _cfg(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


def _graph_pipeline(model, criterion, optimizer, cfg, step_loss, first, clip, extras=0):
    """the GraphPipeline of this (model, optimizer, criterion, batch shape), built from batch `first` at first use"""
    import torch.distributed as tdist
    key = (id(model), id(optimizer), id(criterion), tuple(first["pos"].shape), tuple(sorted(k for k, v in first.items() if torch.is_tensor(v))), extras)
    hit = _PIPELINES.get(key)
    if hit is None:
        dev = first["pos"].device
        main = torch.cuda.Stream(dev)
        main.wait_stream(torch.cuda.current_stream(dev))
        flat, sync_bn = None, False
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            from .dist import FlatGradients
            flat = FlatGradients([p for p in model.parameters() if p.requires_grad], accumulate=False)
            sync_bn = any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
        # the epoch's device-side bookkeeping (main_AA.py:414-416: cm.update, loss_meter.update) as the tail of the captured step
        from . import ops
        v = cfg.num_classes + (1 if cfg.ignore_index is not None else 0)
        book = {"cm": torch.zeros(v, v, dtype=torch.int64, device=dev), "invalid": torch.zeros(1, dtype=torch.int64, device=dev),
                "loss": torch.zeros(1 + extras, dtype=torch.float64, device=dev)}

        def tail(out, data):
            ops.confusion_update(book["cm"], book["invalid"], out["logits"], data["y"], cfg.ignore_index)
            terms = [out["loss"].detach().reshape(1)] + [t.detach().reshape(1) for t in out["parts"]]
            book["loss"].add_(terms[0] if len(terms) == 1 else torch.cat(terms))  # one launch (fp32 -> fp64 inside it)
        if v > 64 or os.environ.get("AMC3D_EAGER_BOOKKEEPING"):  # (ops.confusion_update's histogram holds 64 x 64 bins)
            tail = None
        with torch.cuda.stream(main):
            pipe = GraphPipeline(model, lambda data: step_loss(data, data["y"]), criterion.contrast_head, optimizer, first,
                                 cfg.num_classes, cfg.ignore_index, cfg.ambiguity_args, max_grad_norm=clip, flat_grads=flat,
                                 sync_bn=sync_bn, lanes=int(_cfg(cfg, "fps_lanes", 0) or 0),
                                 tail=tail)
        hit = _PIPELINES[key] = (pipe, main, book)
    return hit


def _run_epoch(model, train_loader, criterion, optimizer, scheduler, scaler, epoch, cfg, prefetch_depth, device, step_loss,
               extras):
    activate()
    from openpoints.utils import ConfusionMatrix
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cm = ConfusionMatrix(num_classes=cfg.num_classes, ignore_index=cfg.ignore_index)
    model.train()
    head = getattr(criterion, "contrast_head", None)
    batches = _to_device_batches(train_loader, cfg, device)
    use_amp = bool(_cfg(cfg, "use_amp", False))
    clip = _cfg(cfg, "grad_norm_clip", None)
    clip = clip if (clip is not None and clip > 0.) else None
    loss_sum = torch.zeros(1 + extras, dtype=torch.float64, device=device)
    n_batches = 0
    graphs_ok = (head is not None and not use_amp and cfg.step_per_update == 1 and _cfg(cfg, "graph_pipeline", True)
                 and not os.environ.get("AMC3D_EAGER_TRAIN")
                 and not isinstance(model, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)))
    if graphs_ok:
        first = next(batches, None)
        if first is None:
            graphs_ok = False
            batches = iter(())
    if graphs_ok:
        pipe, main, book = _graph_pipeline(model, criterion, optimizer, cfg, step_loss, first, clip, extras)
        shape = tuple(first["pos"].shape)
        odd = []  # batches of another shape (a ragged last batch): trained eagerly after the pipeline has drained

        def same_shape(it):
            for b in it:
                if tuple(b["pos"].shape) == shape:
                    yield b
                else:
                    odd.append(b)
        cur = torch.cuda.current_stream(device)
        main.wait_stream(cur)
        with torch.cuda.stream(main):
            in_graph = pipe.tail is not None
            if in_graph:  # the captured step keeps the books (warm-up passes and earlier epochs have written to them)
                book["cm"].zero_()
                book["invalid"].zero_()
                book["loss"].zero_()
            for out in pipe.run(same_shape(itertools.chain([first], batches))):
                if not cfg.sched_on_epoch:
                    scheduler.step(epoch)
                if not in_graph:
                    cm.update_from_logits(out["logits"], out["target"])
                    if extras:
                        loss_sum.add_(torch.stack([out["loss"].detach()] + [v.detach() for v in out["parts"]]))
                    else:
                        loss_sum.add_(out["loss"].detach().reshape(1))  # one launch (the fp32 -> fp64 promotion happens in it)
                n_batches += 1
            if in_graph and n_batches:
                cm.add_counts(book["cm"], book["invalid"])
                loss_sum.add_(book["loss"][:loss_sum.numel()])
        cur.wait_stream(main)
        batches = iter(odd)
    elif head is not None and prefetch_depth > 0 and not use_amp:
        batches = GeometryPrefetcher(batches, model, head, cfg.num_classes, cfg.ignore_index, cfg.ambiguity_args,
                                     depth=prefetch_depth)
    num_iter = 0
    for data in batches:
        num_iter += 1
        target = data["y"]
        with torch.autocast("cuda", enabled=use_amp):
            logits, loss, parts = step_loss(data, target)
        if use_amp:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        if num_iter == cfg.step_per_update:
            # (with use_amp the reference clips the still-scaled gradients, main_AA.py:402-409; kept as it is)
            folded = clip is not None and not use_amp and type(optimizer).__name__ == "FusedAdamW"  # clip inside the optimizer's launch
            if clip is not None and not folded:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip, norm_type=2)
            num_iter = 0
            if use_amp:
                scaler.step(optimizer)
                scaler.update()
            elif folded:
                optimizer.step(max_grad_norm=clip)
            else:
                optimizer.step()
            optimizer.zero_grad()
            if not cfg.sched_on_epoch:
                scheduler.step(epoch)
        cm.update_from_logits(logits, target)
        loss_sum += torch.stack([loss.detach()] + [v.detach() for v in parts]).double()
        n_batches += 1
    miou, macc, oa, ious, accs = cm.all_metrics()
    return tuple((loss_sum / max(1, n_batches)).tolist()) + (miou, macc, oa, ious, accs)
