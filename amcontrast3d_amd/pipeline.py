"""Geometry prefetching for a training loop (the reference's `train_one_epoch`, examples/segmentation/main_AA.py:370-428).

Half of a train step depends only on coordinates and labels: the four FPS levels, ball queries, relative
positions, 3-NN weights and the loss's k-NN / class votes / positive masks / ambiguities
(`amcontrast3d_amd.geometry`).  The FPS chain is a latency-bound kernel that keeps 8 of the 256 CUs busy for
~10 ms; run in line it is half of the step.  `GeometryPrefetcher` wraps the batch iterator of a training loop
and computes that half for the NEXT batches on two side streams while the model and criterion work on the
current one:

    for data in GeometryPrefetcher(batches, model, criterion.contrast_head, num_classes, ignore_index, aargs):
        logits, stage = model(data)                      # finds data['_geometry'], skips FPS / ball query / 3-NN
        loss = criterion(logits, data['y'], stage, num_classes, ignore_index, aargs)   # finds the loss geometry
        ...

The batches yielded are the caller's dicts (pos, y already on the GPU; 'x' etc. untouched) with one extra key.
Results are identical to the un-prefetched path: the same kernels on the same inputs, only earlier and on another
stream (tests/test_gpu_pipeline.py).  bench.py implements the same schedule with hipGraphs and static buffers.

The two side streams are picked from PyTorch's pool so that they and the caller's stream sit on three different
hardware queues (probed once per process, _side_streams): ordinary HIP streams share four queues round-robin, and a
stream that lands on the queue of a running FPS kernel waits milliseconds for it.
"""
import collections

import torch

from . import geometry


_streams = {}  # device -> (FPS stream, geometry stream): shared by every prefetcher of the process


def _shares_queue(a, b, cycles=3_000_000):
    """True when work on stream `b` has to wait for work on stream `a`: both were mapped onto the same hardware queue.
    Probe: a ~1.5 ms spin kernel on `a`, a tiny kernel on `b`, and the host time until `b` is idle."""
    import time
    torch.cuda.synchronize()
    with torch.cuda.stream(b):
        probe = torch.zeros(8, device=f"cuda:{b.device_index}")
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    t0 = time.perf_counter()
    with torch.cuda.stream(b):
        probe.add_(1)
    b.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt > 0.5e-3


def _side_streams(dev):
    """One pair of side streams per device, however many prefetchers (one per epoch) come and go -- the caching
    allocator keeps a memory pool per stream, so a fresh pair per epoch would strand the previous epoch's blocks
    (measured: +0.4 GiB reserved per epoch) -- chosen so that the caller's stream and the two side streams sit on three
    different hardware queues.  Ordinary HIP streams share four queues per process, handed out round-robin; a stream
    that lands on the queue of the FPS stream waits ~10 ms per step for it, and which one does depends on every stream
    created before.  Streams with a queue of their own (ops.dedicated_stream, what bench.py's
    captured pipeline uses) are not an option for this eager loop: launched kernel by kernel they lose the overlap
    altogether (measured: 21 ms/step against 9.8 with pooled streams), so the pool is probed instead."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _streams:
        cur = torch.cuda.current_stream(dev)
        chosen = []
        with torch.cuda.device(dev):
            for _ in range(12):  # successive pool streams cycle through the hardware queues
                cand = torch.cuda.Stream(dev)
                if not any(_shares_queue(r, cand) for r in [cur] + chosen):
                    chosen.append(cand)
                if len(chosen) == 2:
                    break
            while len(chosen) < 2:  # fewer than three free queues (GPU_MAX_HW_QUEUES < 3): overlap what can be overlapped
                chosen.append(torch.cuda.Stream(dev))
        _streams[key] = tuple(chosen)
    return _streams[key]


class GeometryPrefetcher:
    """Iterate `batches` (dicts with 'pos' (B,N,3) and 'y' (B,N) on the GPU), `depth` batches ahead of the
    consumer: FPS chains on one side stream, everything hanging off them on a second one."""

    def __init__(self, batches, model, contrast_head, num_classes, ignore_index, ambiguity_args, depth=2):
        assert depth >= 1
        self.it = iter(batches)
        self.model, self.head = model, contrast_head
        self.num_classes, self.ignore_index, self.aargs = num_classes, ignore_index, ambiguity_args
        self.depth = depth
        self.queue = collections.deque()
        self.s_fps = self.s_rest = None
        self.exhausted = False

    def __iter__(self):
        return self

    def _launch(self, data):
        dev = data["pos"].device
        if self.s_fps is None:
            self.s_fps, self.s_rest = _side_streams(dev)
        cur = torch.cuda.current_stream(dev)
        self.s_fps.wait_stream(cur)  # pos / y were produced (copied to the GPU) on the caller's stream
        with torch.cuda.stream(self.s_fps):
            fps = geometry.precompute_fps(self.model, data)
        self.s_rest.wait_stream(self.s_fps)
        with torch.cuda.stream(self.s_rest):
            plan = geometry.precompute_rest(self.model, self.head, data, fps, self.num_classes, self.ignore_index,
                                            self.aargs)
            done = torch.cuda.Event()
            done.record()
        self.queue.append((data, plan, done))

    def _fill(self):
        while not self.exhausted and len(self.queue) < self.depth + 1:
            try:
                data = next(self.it)
            except StopIteration:
                self.exhausted = True
                return
            self._launch(data)

    def __next__(self):
        self._fill()
        if not self.queue:
            raise StopIteration
        data, plan, done = self.queue.popleft()
        cur = torch.cuda.current_stream(data["pos"].device)
        cur.wait_event(done)
        # the plan's tensors were allocated on the side streams and are consumed on the caller's: tell the caching
        # allocator, so that their memory is not handed out again while the caller's kernels still read it
        geometry._walk(plan, lambda t: t.record_stream(cur) if t.is_cuda else None)
        data["_geometry"] = plan
        self._fill()  # launch the geometry of a later batch before the caller's model call is enqueued
        return data
