"""hipGraph capture of a train step that contains collectives.

Round 3: with RCCL the collectives are captured INTO the graph (collectives_capturable(); torch.distributed's nccl backend
issues them as kernels on the caller's stream, which a stream capture records like any other launch) -- the feature half stays
one graph however many SyncBatchNorm layers it has.  What follows is the form for backends that cannot be captured (gloo
rehearsals) and the fallback: a chain of graphs with the collectives between them.

Why.  At world_size > 1 the reference turns every BatchNorm into SyncBatchNorm (examples/segmentation/main_AA.py:146-148,
820): 34 layers x (one statistics all-reduce forward, one backward) sit in the middle of the feature half of a step.
Launching that half kernel by kernel is launch-bound (~700 launches); capturing it as ONE hipGraph would put RCCL
collectives inside a graph.  `SegmentedGraph` does neither: the step function runs once under stream capture, and every
time it reaches a collective (`collective(fn)`, called by ops.SyncBatchNormFused) the capture is ended, `fn` is kept as
an eager call, and a new capture begins in the same memory pool.  Replay = graph, collective, graph, collective, ...
in the recorded order on the current stream: the kernels are replayed as graphs, the collectives are ordinary
`torch.distributed` calls between them, as they would be in an eager loop.  (Collectives issued while graphs are still
being captured -- warm-up steps, the capturing pass itself, barriers -- go through `on_side_stream`; before capturing on a
stream that has run collectives, call `quiesce()`.)

Backward.  autograd normally runs CUDA nodes on a worker thread, and a stream capture must be ended by the thread that
began it; `capture()` therefore runs the function under `torch.autograd.set_multithreading_enabled(False)`, which keeps
backward on the calling thread.

During the capturing pass the collectives are executed too (on memory whose producing kernels were captured, not run:
the values are meaningless, the call sequence is what matters): every rank issues the same collectives in the same
order in that pass as in every replay.
"""
import torch

_active = None  # the SegmentedGraph that is capturing on this thread, if any


_side_streams = {}  # device index -> the stream every collective of this process is issued on
captured_collectives = 0  # collectives recorded INTO graphs so far (GraphPipeline reads the difference around a capture)


def collectives_capturable():
    """True when this process's default group is RCCL (nccl backend): its collectives are ordinary kernels on the caller's
    stream and can be recorded into a hipGraph.  gloo (the CPU rehearsals) cannot; AMC3D_SEGMENTED_COLLECTIVES=1 keeps the
    round-2 form (SegmentedGraph: the graph is cut at every collective) for RCCL too."""
    import os
    import torch.distributed as dist
    return (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
            and not os.environ.get("AMC3D_SEGMENTED_COLLECTIVES"))


def on_side_stream(fn):
    """Run `fn` (a torch.distributed call) on this device's collective stream, ordered after the current stream's work and
    before its later work.  Why not on the current stream: c10d runs a blocking collective on the caller's stream and
    records the work's completion event there; its watchdog thread polls that event until it sees it complete -- up to
    ~100 ms later -- and HIP refuses to query an event whose stream is capturing AT THAT MOMENT (hipErrorCapturedEvent, which
    also invalidates the capture).  A training loop that captures graphs on its main stream after any eager collective on
    that stream (warm-up steps, a barrier) therefore aborts every few runs.  The collective stream never captures."""
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return fn()  # CPU process groups (gloo tests)
    cur = torch.cuda.current_stream()
    if torch.cuda.is_current_stream_capturing():
        # a plain capture: the collective becomes a node of the graph (RCCL supports stream capture; what
        # pipeline.GraphPipeline does with the nccl backend since round 3 -- no graph cut, no eager launch per collective)
        global captured_collectives
        captured_collectives += 1
        return fn()
    side = _side_streams.get(cur.device.index)
    if side is None:
        side = _side_streams[cur.device.index] = torch.cuda.Stream(device=cur.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = fn()
    cur.wait_stream(side)
    return out


def quiesce(seconds=0.3):
    """Call before capturing graphs on a stream that has run collectives: drains the GPU and gives c10d's watchdog thread
    (poll interval ~100 ms) time to retire the finished works, whose events it must not query once the stream captures."""
    import time
    torch.cuda.synchronize()
    time.sleep(seconds)


def collective(fn):
    """Run `fn` (a torch.distributed call) now, on the collective stream; inside SegmentedGraph.capture() also cut the
    graph here."""
    if _active is not None:
        return _active._cut(fn)
    return on_side_stream(fn)


class SegmentedGraph:
    def __init__(self, capture_error_mode="thread_local"):
        self.mode = capture_error_mode
        self.items = []      # torch.cuda.CUDAGraph | callable, in replay order
        self.pool = None
        self._cur = None

    # -- capture ---------------------------------------------------------------------------------
    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self.pool, capture_error_mode=self.mode)

    def _end(self):
        self._cur.capture_end()
        self.items.append(self._cur)
        self._cur = None

    def _cut(self, fn):
        self._end()
        self.items.append(fn)
        # keeps the ranks' collective sequences aligned during the capturing pass (on the collective stream: on_side_stream)
        out = on_side_stream(fn)
        self._begin()
        return out

    def capture(self, fn, stream=None):
        """Capture fn() -- forward, loss, backward, anything -- on `stream` (default: the current stream, which must
        not be the legacy default stream).  Collectives inside must go through graphs.collective()."""
        global _active
        assert _active is None and not self.items, "one capture per SegmentedGraph"
        stream = stream if stream is not None else torch.cuda.current_stream()
        self.pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
            _active = self
            try:
                self._begin()
                fn()
                self._end()
            finally:
                _active = None
                if self._cur is not None:  # fn raised inside a capture: leave the stream usable
                    try:
                        self._cur.capture_end()
                    except Exception:
                        pass
                    self._cur = None
        torch.cuda.synchronize()
        return self

    # -- replay ----------------------------------------------------------------------------------
    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()  # on the current stream: see quiesce() before capturing on this stream again

    @property
    def segments(self):
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)

    @property
    def collectives(self):
        return len(self.items) - self.segments


# ---- what a captured graph is made of ---------------------------------------------------------------------------------
_NODE_TYPES = ("kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record", "ext_semaphore_signal",
               "ext_semaphore_wait", "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "batch_mem_op")


def node_type_counts(graph):
    """{node type: count} of a captured torch.cuda.CUDAGraph made with keep_graph=True (hipGraphGetNodes / hipGraphNodeGetType;
    child graphs are not descended into).  What it is for: on ROCm 7.2 a MEMSET node is not reliably ordered before the kernel
    nodes behind it at replay -- rocPRIM's radix sort and torch's multi-block reductions both zero their
    counters with hipMemsetAsync -- so the product's graphs must hold none (GraphPipeline(audit=True), tests)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    if n.value and hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    counts = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        name = _NODE_TYPES[t.value] if 0 <= t.value < len(_NODE_TYPES) else f"type_{t.value}"
        counts[name] = counts.get(name, 0) + 1
    return counts
