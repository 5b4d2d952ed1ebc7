"""One process per GPU, plain data parallel over RCCL (torch.distributed backend 'nccl').

The path shards by scene: every rank trains on its own clouds, the only
exchange is the gradient all-reduce (+ SyncBatchNorm statistics, which the reference forces on
whenever world_size > 1: examples/segmentation/main_AA.py:146-148, 820).  Helpers here are shared by
bench.py and the gloo tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_from_env(backend=None):
    """Initialise the default process group when launched with WORLD_SIZE > 1; returns (rank, local, world)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner to STDOUT under NCCL_DEBUG=VERSION/INFO; rank 0's stdout is the one JSON line.
        # (A value the user set is kept.)
        os.environ.setdefault("NCCL_DEBUG", os.environ.get("AMC3D_NCCL_DEBUG", "WARN"))
        if backend is None:  # AMC3D_DIST_BACKEND=gloo: rehearse the multi-rank control flow on a one-GPU box
            backend = os.environ.get("AMC3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def scene_ids(rank, world, per_rank, step=0):
    """Scene ids of one rank for one step: disjoint across ranks (DistributedSampler semantics,
    dataset/build.py:78-88), fixed per-rank batch -> weak scaling."""
    base = (step * world + rank) * per_rank
    return list(range(base, base + per_rank))


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (step time = slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    from .graphs import on_side_stream
    t = torch.tensor([value], dtype=torch.float64, device=device)
    on_side_stream(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .graphs import on_side_stream
        on_side_stream(dist.barrier)


def wrap_data_parallel(model, device, world):
    """SyncBN conversion + DistributedDataParallel, as main_AA.py:146-152 does for world_size > 1."""
    if world <= 1:
        return model
    if device.type == "cuda":
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], output_device=device.index,
                                                         gradient_as_bucket_view=True)
    return torch.nn.parallel.DistributedDataParallel(model)


def allreduce_gradients(params, bucket_bytes=32 << 20):
    """Average gradients across ranks in flat buckets (for modules that bypass DDP's hooks).
    Buckets are filled in parameter order; one all-reduce per bucket."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nbytes
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        from .graphs import on_side_stream
        on_side_stream(lambda: dist.all_reduce(flat))
        flat.div_(world)
        off = 0
        for g in b:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)


class FlatGradients:
    """Every parameter's gradient in ONE buffer, so that the data-parallel exchange of a step is a single all-reduce
    on memory the optimizer reads directly: no bucket assembly and no copy-back (allreduce_gradients needs ~2 launches
    per parameter for those, ~0.7 ms of eager launches per PointNeXt-S step during which the GPU idles between the
    captured backward and the captured optimizer step).

    Two ways to get the gradients there:
      accumulate=True   .grad is a view of the buffer from the start; autograd accumulates into an existing .grad in
                        place, zero() replaces optimizer.zero_grad().  One `add_` per parameter and step (0.17 ms).
      accumulate=False  backward runs with .grad = None (autograd just keeps the tensors it produced); gather() then
                        copies them into the buffer with one multi-tensor copy and points .grad at the views, which
                        is what the optimizer (and a graph capturing it) reads; release() sets .grad = None again
                        before the next backward.  Under graph capture gather() is part of the captured half.
    """

    def __init__(self, params, accumulate=True):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and len({(p.device, p.dtype) for p in self.params}) == 1
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=self.params[0].dtype, device=self.params[0].device)
        self.accumulate = accumulate
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if accumulate:
            for p, v in zip(self.params, self.views):
                p.grad = v

    def zero(self):
        """start of a step: accumulate mode clears the buffer (a kernel, not a memset -- hipMemset nodes race under
        graph replay on ROCm 7.2); copy mode hands autograd empty .grad slots"""
        if self.accumulate:
            self.flat.fill_(0)
        else:
            self.release()

    def release(self):
        for p in self.params:
            p.grad = None

    def gather(self):
        """copy mode, after backward: fresh gradients -> buffer (one multi-tensor copy), .grad -> views"""
        if self.accumulate:
            return
        have = [(v, p.grad) for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for p, v in zip(self.params, self.views) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v in missing:  # a parameter that took no part in this backward
            v.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def intact(self):
        """the views are in place (after gather() in copy mode): nothing replaced a .grad behind our back"""
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def allreduce(self):
        """average over ranks: one collective (also issued in a one-rank group, so that a single-GPU rehearsal
        exercises the same RCCL call)"""
        if not (dist.is_available() and dist.is_initialized()):
            return
        from .graphs import on_side_stream  # never on a stream that captures graphs (graphs.on_side_stream)
        if dist.get_backend() == "nccl":
            on_side_stream(lambda: dist.all_reduce(self.flat, op=dist.ReduceOp.AVG))
        else:  # gloo has no AVG
            on_side_stream(lambda: dist.all_reduce(self.flat))
            self.flat.div_(dist.get_world_size())
