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

Both side streams own their hardware queue (ops.dedicated_stream).  Ordinary HIP streams of a process share four
queues round-robin, and a stream that lands on the queue of a running FPS kernel waits milliseconds for it -- the
training stream, a graph's internal branch or RCCL's, depending on how many streams were created before (measured:
the same loop at 10.4 or 14 ms/step with and without a process group alive).
"""
import collections

import torch

from . import geometry


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
            from . import ops
            self.s_fps, self.s_rest = ops.dedicated_stream(dev), ops.dedicated_stream(dev)
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
