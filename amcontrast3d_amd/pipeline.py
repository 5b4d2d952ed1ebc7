"""The pipelined train step of the hot path (the reference's `train_one_epoch`, examples/segmentation/main_AA.py:370-428).

Two schedulers of the same computation live here:

    GraphPipeline        hipGraph replays on three hardware queues -- what `train.train_one_epoch` and `bench.py` run
    GeometryPrefetcher   the eager form (kernel-by-kernel launches on pooled streams) -- the fallback for loops the
                         graphs cannot express (gradient accumulation, GradScaler, DDP-wrapped models, ragged batches)

Both rest on the geometry / feature split (amcontrast3d_amd/geometry.py).  GeometryPrefetcher first:

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


_queues = {}  # (device, CU mask of the geometry queue) -> (sampling queue, geometry queue)


def _dedicated_queues(dev, geometry_cus):
    """One pair of dedicated hardware queues per device and CU mask for every GraphPipeline of the process (pipelines of one
    process never run at the same time).  The runtime schedules four hardware queues natively; with a pair per pipeline a
    second pipeline would be the fifth and sixth queue, and the very same step then takes 17 ms instead of 7 (measured in
    round 2 with one queue per FPS lane)."""
    from . import ops
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(geometry_cus))
    if key not in _queues:
        fps = next((q[0] for k, q in _queues.items() if k[0] == key[0]), None) or ops.dedicated_stream(dev)
        _queues[key] = (fps, ops.dedicated_stream(dev, 0, geometry_cus))
    return _queues[key]


# ---------------------------------------------------------------------------------------------------------------------
class GraphPipeline:
    """A software pipeline of hipGraphs over successive batches of fixed shape:

        sampling queue   ALL FPS levels of J future batches as ONE launch every J ticks (a workgroup per cloud: the chain
                         is latency-bound, more clouds per launch cost nothing); two J-batch buffers used in turn
        geometry queue   hand-down copies, then the neighbourhood + loss geometry (ball queries, relative positions,
                         reverse edge lists, 3-NN, the loss's k-NN / votes / masks / ambiguities / anchor lists) of the
                         batch that trains NEXT tick; CU-masked; three captured variants filling three result sets in turn
        main stream      features of the current batch: forward + loss + backward (+ gradient all-reduce) + clip + optimizer
                         step; three captured variants, variant v reading the very input set and result set the geometry
                         variant v worked on one tick earlier -- nothing is copied on the main stream between two steps, and
                         it records no event that another queue waits for (a set is refilled two ticks after it was read,
                         which the host checks; with two sets the geometry queue had to wait for the training stream's event,
                         0.21 ms per step on the training stream: tools/bubble_probe.py)

    Every batch goes through exactly the computation of the eager loop, once; the pipeline decides WHEN its coordinate-only
    half runs (1 .. 2J ticks ahead).  Batches come out in the order they went in.

        pipe = GraphPipeline(model, step_loss, head, optimizer, example, num_classes, ignore_index, ambiguity_args)
        for out in pipe.run(batches):       # batches: iterable of dicts of DEVICE tensors shaped like `example`
            out["loss"], out["logits"], out["target"], out["parts"]   # static tensors, valid until the next-but-one batch

    `step_loss(data) -> (logits, loss, parts)` is the model + criterion call (parts: extra scalars to report);
    `head` = criterion.contrast_head.  The optimizer step is captured when the optimizer can be (FusedAdamW, or torch's
    capturable ones), else it runs eagerly after the feature graph.  Building the pipeline runs three warm-up steps on
    `example`; parameters, buffers and optimizer state are restored afterwards (`keep_state=True`) so that training starts
    from the state it was given.  With `flat_grads` (N > 1) the gradient exchange is one all-reduce between the feature
    graph and the update; SyncBatchNorm layers cut the feature graph at their collectives (graphs.SegmentedGraph).
    The current stream at construction must not be the legacy default stream (graphs are captured on it)."""

    def __init__(self, model, step_loss, head, optimizer, example, num_classes, ignore_index, ambiguity_args, *,
                 lanes=0, max_grad_norm=None, flat_grads=None, sync_bn=False, keep_state=True, geometry_cus=None,
                 amp_dtype=None, verbose=False, tail=None, audit=None):
        from . import ops
        self.model, self.step_loss, self.head, self.opt = model, step_loss, head, optimizer
        # tail(out, data): the caller's per-iteration bookkeeping on device (confusion matrix, loss sums: train.py) recorded as the
        # tail of the feature graph, into tensors the caller owns -- eager launches between two replays cost the training stream
        # more than their kernels (0.11 ms per step for two small ones).  It also runs in the warm-up passes: reset afterwards.
        self.tail = tail
        # audit: keep every captured graph's node list and record its node types in self.graph_nodes {name: [counts per graph]}
        # (graphs.node_type_counts) -- the product's graphs must hold no memset node; AMC3D_AUDIT_GRAPHS=1 turns it on and raises
        import os as _os
        self.audit = bool(_os.environ.get("AMC3D_AUDIT_GRAPHS")) if audit is None else bool(audit)
        self.graph_nodes = {}
        self.ncls, self.ignore, self.aargs = num_classes, ignore_index, ambiguity_args
        self.clip, self.flatg, self.sync_bn, self.amp_dtype = max_grad_norm, flat_grads, sync_bn, amp_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.dev = example["pos"].device
        self.B, self.N = example["pos"].shape[:2]
        self.keys = [k for k, v in example.items() if torch.is_tensor(v)]
        self.main = torch.cuda.current_stream(self.dev)
        assert self.main != torch.cuda.default_stream(self.dev), "GraphPipeline captures on the current stream: make a side stream current"
        self.nlevels = len(list(geometry._unwrap(model).encoder.encoder))
        snapshot = self._snapshot() if keep_state else None
        self.lanes = J = max(2, lanes if lanes > 0 else self._choose_lanes(example, verbose))
        # hardware queues of their own for the two background chains (ordinary streams share four queues round-robin and
        # whatever shares a queue with a running FPS kernel waits milliseconds for it); the geometry queue is confined
        # to 11/16 of the CUs for clouds the register-resident FPS kernel handles (<= 24576 points; measured: its kernels
        # are background work with slack and slow the feature half more than they gain when they spread over the chip)
        ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        if geometry_cus is None:
            import os
            # (sweeps.  End of round 3, feature graph 4.84 ms: 9/16 6.47 ms -- the next batch's geometry becomes the critical path --
            # 10/16 6.02-6.03, 11/16 6.03, 12/16 6.04: one sixteenth of margin to that cliff for a box whose geometry runs slower)
            sixteenths = int(os.environ.get("AMC3D_GEO_CUS_16THS", "11"))
            geometry_cus = sixteenths * ncu // 16 if self.N <= 24576 else 0
        self.s_fps, self.s_geo = _dedicated_queues(self.dev, geometry_cus)
        self.geometry_cus = geometry_cus
        self.ev_lane = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_geo, self.ev_rot = torch.cuda.Event(), torch.cuda.Event()
        # "tick t's feature graph has finished": recorded on the training stream, only ever queried by the HOST.  The geometry
        # queue refills a set two ticks after it was read (three sets in turn), so that this check can be the host's: an event
        # of the training stream that ANOTHER QUEUE waits for costs the training stream 0.21 ms per step (tools/bubble_probe.py)
        from . import schedule
        self.ev_done = [torch.cuda.Event() for _ in range(schedule.SETS)]
        self._done_recorded = [False] * schedule.SETS
        self._build(example)
        if snapshot is not None:
            self._restore(snapshot)
        self.tick = 0
        self._set_valid = [False] * schedule.SETS
        self._lane_valid = [[False] * J, [False] * J]
        self._lr = tuple(g["lr"] for g in optimizer.param_groups)

    # -- pieces of a step --------------------------------------------------------------------------------------------
    def _fps_all(self, batch):
        return geometry.precompute_fps_levels(self.model, batch["pos"], 0, self.nlevels)

    def _rest(self, batch, fps):
        return geometry.precompute_rest(self.model, self.head, batch, fps, self.ncls, self.ignore, self.aargs)

    def _fwd_bwd(self, data, out):
        if self.flatg is not None:
            self.flatg.zero()
        with torch.autocast("cuda", dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None):
            logits, loss, parts = self.step_loss(data)
        loss.backward()
        if self.flatg is not None:
            self.flatg.gather()
        out.update(logits=logits, loss=loss, parts=parts)
        if self.tail is not None:
            with torch.no_grad():
                self.tail(out, data)

    def _update(self):
        if type(self.opt).__name__ == "FusedAdamW":  # clip_grad_norm_ + AdamW as two launches (csrc/optim.hip)
            self.opt.step(max_grad_norm=self.clip)
        else:
            if self.clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.clip, norm_type=2)
            self.opt.step()

    def _copy_batch(self, dst, src):
        torch._foreach_copy_([dst[k] for k in self.keys], [src[k] for k in self.keys])

    def _ms(self, fn, reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def _choose_lanes(self, example, verbose):
        """batches per joint FPS launch: 4 where the first-level chain is about a feature half long (24k-point clouds: the
        launch then runs 10 ms every fourth tick); for long chains (64k / 120k-point clouds in batches of 1-2) as many as
        make the chain of ALL levels fit into J ticks next to a busy chip, at most 24"""
        data = dict(example)
        t_fps = self._ms(lambda: geometry.precompute_fps_levels(self.model, data["pos"], 0, 2), 1)
        data["_geometry"] = geometry.precompute(self.model, self.head, data, self.ncls, self.ignore, self.aargs)
        out = {}

        def feat():
            self._fwd_bwd(data, out)
            if self.flatg is None:
                for p in self.params:
                    p.grad = None
        t_feat = self._ms(feat, 2)
        t_all = None
        if t_fps <= 1.5 * t_feat:
            lanes = 4
        else:
            # The chain runs ~2.7 x slower next to a busy chip than alone and the captured step takes ~0.8 of the eager pass: it
            # fits into J steps from J ~ t_all / (0.28 t_feat).  Measured over whole launch periods (a window that is not a multiple
            # of J counts the joint launches unevenly and moves the average by up to 5 %): XL + ++ at 2 x 64000 -- 7 lanes 19.1 ms
            # per step, 8: 18.8, 9: 18.55, 10: 18.54, 11: 18.33; 1 x 120000 bf16 (the L2-resident kernel, slower still as more
            # clouds share the L2) -- 12 lanes 19.1, 16: 17.5, 20: 17.4.
            t_all = self._ms(lambda: self._fps_all(data), 1)
            lanes = int(-(-t_all // max(0.28 * t_feat, 1e-3)))
            if self.N >= 100000:
                lanes = max(lanes, 24)
            lanes = int(min(24, max(3, lanes)))
        if verbose:
            import sys
            print(f"GraphPipeline: sampling chain {t_fps:.1f} ms (first level{'' if t_all is None else f', {t_all:.1f} all levels'}), "
                  f"eager feature half {t_feat:.1f} ms -> {lanes} batches per joint FPS launch", file=sys.stderr)
        return lanes

    # -- state kept across the warm-up --------------------------------------------------------------------------------
    def _snapshot(self):
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        ost = {id(t): t.detach().clone() for st in self.opt.state.values() for t in st.values() if torch.is_tensor(t)}
        return sd, ost, torch.cuda.get_rng_state(self.dev)

    def _restore(self, snapshot):
        sd, ost, rng = snapshot
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(sd[k])
            for st in self.opt.state.values():  # in place: the captured update reads these tensors
                for t in st.values():
                    if torch.is_tensor(t):
                        t.copy_(ost[id(t)]) if id(t) in ost else t.zero_()
        torch.cuda.set_rng_state(rng, self.dev)

    # -- capture ------------------------------------------------------------------------------------------------------
    def _build(self, example):
        from .graphs import SegmentedGraph, quiesce
        import os
        import torch.distributed as tdist
        J, B = self.lanes, self.B
        ex = {k: example[k] for k in self.keys}
        lane = lambda t, l: t[l * B:(l + 1) * B]
        # two J-batch buffers of the sampling queue and their outputs (all levels), with per-lane views
        self.in_J = [{k: torch.cat([v] * J) for k, v in ex.items()} for _ in range(2)]
        self.fps_J = [self._fps_all(self.in_J[j]) for j in range(2)]
        self.in_lane = [[{k: lane(v, l) for k, v in self.in_J[j].items()} for l in range(J)] for j in range(2)]
        self.fps_lane = [[geometry._walk(self.fps_J[j], lambda t, l=l: lane(t, l)) for l in range(J)] for j in range(2)]
        # three input sets (batch + its FPS picks) shared by geometry variant v and, one tick later, feature variant v
        from . import schedule
        S = schedule.SETS
        self.set_in = [{k: v.clone() for k, v in ex.items()} for _ in range(S)]
        self.set_fps = [geometry.clone(self.fps_lane[0][0]) for _ in range(S)]
        self.out = [{} for _ in range(S)]
        data0 = dict(self.set_in[0])

        def eager_step():  # everything once, in line: allocator warm-up, lazy initialisations, optimizer state
            data0["_geometry"] = self._rest(self.set_in[0], self._fps_all(self.set_in[0]))
            if self.flatg is None:
                self.opt.zero_grad(set_to_none=True)
            self._fwd_bwd(data0, self.out[0])
            if self.flatg is not None:
                self.flatg.allreduce()
            self._update()
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        has_grad = [p.grad is not None for p in self.params]
        dist_on = tdist.is_available() and tdist.is_initialized()
        if dist_on:
            quiesce()  # c10d's watchdog must not poll an event of a stream that is capturing
        if self.flatg is None:
            self.opt.zero_grad(set_to_none=True)
        mode = "thread_local" if dist_on else "global"
        self.mode = mode
        audit = self.audit

        def G():
            return torch.cuda.CUDAGraph(keep_graph=True) if audit else torch.cuda.CUDAGraph()

        def audited(g, name):
            if not audit:
                return
            counts = _graphs_mod.node_type_counts(g)
            self.graph_nodes.setdefault(name, []).append(counts)
            g.instantiate()
            if counts.get("memset") and os.environ.get("AMC3D_AUDIT_GRAPHS"):
                raise RuntimeError(f"GraphPipeline: the {name} graph holds {counts['memset']} memset node(s): {counts}")
        from . import graphs as _graphs_mod
        # geometry variants: their own output tensors are the two result sets
        self.g_geo, self.rest = [G() for _ in range(S)], []
        for v in range(S):
            with torch.cuda.graph(self.g_geo[v], stream=self.s_geo, capture_error_mode=mode):
                self.rest.append(geometry.split(self._rest(self.set_in[v], self.set_fps[v]))[1])
            audited(self.g_geo[v], "geometry")
        for v, r in enumerate(self.rest):  # a result set may only alias the inputs of its own variant
            other = set()
            geometry._walk([[self.set_fps[u], self.set_in[u]] for u in range(S) if u != v] + [self.in_J, self.fps_J],
                           lambda t: other.add(t.untyped_storage().data_ptr()))

            def check(t):
                assert t.untyped_storage().data_ptr() not in other, "the geometry plan aliases another batch's buffers"
            geometry._walk(r, check)
        # One rank, nothing between backward and update (no gradient all-reduce, no SyncBatchNorm cuts): the update is the tail
        # of both feature graphs -- one graph launch per step instead of two (the launch of the update graph and the gap in front
        # of it were ~0.1 ms of the 0.25 ms the main stream idled per step)
        fused = type(self.opt).__name__ == "FusedAdamW"
        from . import graphs as _graphs
        # RCCL collectives (SyncBatchNorm statistics, the gradient all-reduce) are recorded into the graphs; other backends
        # (gloo rehearsals) cut the feature graph at every collective and issue the gradient all-reduce between two graphs
        self.collectives_captured = dist_on and _graphs.collectives_capturable()
        inline = self.collectives_captured or (self.flatg is None and not self.sync_bn)
        self.update_in_feature_graph = (inline and not os.environ.get("AMC3D_SEPARATE_UPDATE")
                                        and (fused or all(g.get("capturable", False) for g in self.opt.param_groups)))
        n_coll0 = _graphs.captured_collectives
        # feature variants.  Both deliver their gradients in ONE set of static .grad tensors (what the update reads, captured or
        # not): backward runs with .grad = None -- captured with .grad set, autograd would ACCUMULATE into it, last step's
        # gradient plus this one's -- and ends with one multi-tensor copy into the static tensors (3 MB for PointNeXt-S)
        static = None
        if self.flatg is None:
            static = [torch.zeros_like(p) if h else None for p, h in zip(self.params, has_grad)]  # (who gets one: the warm-up steps)
            for p, g0 in zip(self.params, static):
                p.grad = g0
            if fused:
                self.opt.prepare()  # the tensor table of the update, built outside the capture
        self.g_feat = []
        for v in range(S):
            data = dict(self.set_in[v])
            data["_geometry"] = geometry.join(self.set_fps[v], self.rest[v])

            def body(data=data, v=v):
                if static is not None:
                    for p in self.params:
                        p.grad = None
                self._fwd_bwd(data, self.out[v])
                if static is not None:
                    assert all((g0 is None) == (p.grad is None) for g0, p in zip(static, self.params)), \
                        "a parameter's gradient appeared / vanished between the warm-up and the capture"
                    pairs = [(g0, p.grad) for g0, p in zip(static, self.params) if g0 is not None]
                    torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
                    for p, g0 in zip(self.params, static):
                        p.grad = g0
                if self.flatg is not None and self.collectives_captured:
                    self.flatg.allreduce()  # recorded into the graph
                if self.update_in_feature_graph:
                    self._update()
            if self.sync_bn and not self.collectives_captured:
                # the statistics all-reduces cannot be captured: a chain of graphs with eager collectives between
                g = SegmentedGraph(mode).capture(body, stream=self.main)
            else:
                g = G()
                kw = {"pool": self.g_feat[0].pool()} if v else {}  # the variants never run at the same time
                with torch.cuda.graph(g, stream=self.main, capture_error_mode=mode, **kw):
                    body()
                audited(g, "features")
            self.g_feat.append(g)
            if v == 0:
                self.collectives_in_graph = _graphs.captured_collectives - n_coll0
        # the update: captured where the optimizer allows it
        self.g_update = None
        if self.update_in_feature_graph:
            pass
        elif fused or all(g.get("capturable", False) for g in self.opt.param_groups):
            if fused:
                self.opt.prepare()
            self.g_update = G()
            with torch.cuda.graph(self.g_update, stream=self.main, capture_error_mode=mode):
                self._update()
            audited(self.g_update, "update")
        # hand-down on the geometry queue, one graph per tick of the period 2J: lane (buffer jc, lane l) -> input set v1
        self.g_side = []
        for t in range(schedule.period(J)):
            plan = schedule.tick_plan(t, J)
            (jc, l), v1 = plan["consume"], plan["fill"]
            g = G()
            with torch.cuda.graph(g, stream=self.s_geo, capture_error_mode=mode):
                geometry.copy_into(self.set_fps[v1], self.fps_lane[jc][l])
                self._copy_batch(self.set_in[v1], self.in_lane[jc][l])
            audited(g, "hand_down")
            self.g_side.append(g)
        self.g_fps = [G(), G()]
        for j in range(2):
            with torch.cuda.graph(self.g_fps[j], stream=self.s_fps, capture_error_mode=mode):
                geometry.copy_into(self.fps_J[j], self._fps_all(self.in_J[j]))
            audited(self.g_fps[j], "sampling")
        torch.cuda.synchronize()

    # -- running ------------------------------------------------------------------------------------------------------
    @property
    def depth(self):
        """ticks between a batch entering the pipeline and its train step: J + 1 for the first lane of a launch .. 2J"""
        return 2 * self.lanes

    def _critical_path(self, v0):
        captured = self.g_update is not None or self.update_in_feature_graph
        if captured:
            lr = tuple(g["lr"] for g in self.opt.param_groups)
            if lr != self._lr:  # a scheduler stepped: the captured update reads its learning rates from device memory
                if hasattr(self.opt, "sync_hyperparameters"):
                    self.opt.sync_hyperparameters()
                self._lr = lr
        self.g_feat[v0].replay()
        if self.update_in_feature_graph:
            return
        if self.flatg is not None and not self.collectives_captured:
            self.flatg.allreduce()
        if self.g_update is not None:
            self.g_update.replay()
        else:
            self._update()

    def _tick(self, it):
        """one tick: train the batch in input set v0 (if it holds one), prepare the next one's geometry, every J ticks
        load J new batches and launch their sampling.  -> (result dict or None, pipeline still holds batches)"""
        from . import schedule
        import os
        # diagnostic: leave parts out once every buffer holds results (they go stale; timing only, with one resident batch)
        skip = os.environ.get("AMC3D_PIPE_SKIP", "") if self.tick > 6 * self.lanes else ""
        J, t = self.lanes, self.tick % schedule.period(self.lanes)
        self.tick += 1
        plan = schedule.tick_plan(t, J)
        v0, v1, (jc, l), jl = plan["train"], plan["fill"], plan["consume"], plan["launch"]
        out = None
        # the main stream goes first: the side launches below take the host 0.3-0.5 ms
        self.main.wait_event(self.ev_geo)   # result set v0 is complete
        if self._set_valid[v0]:
            self._critical_path(v0)
            out = dict(self.out[v0], target=self.set_in[v0]["y"], data=self.set_in[v0])
            self.ev_done[v0].record(self.main)
            self._done_recorded[v0] = True
        # set v1 is refilled now; the feature graph that read it last ran two ticks ago: the host makes sure it has finished
        # (it has, unless the host is more than two steps ahead of the GPU) -- no device-side wait for the training stream
        if self._done_recorded[v1]:
            self.ev_done[v1].synchronize()
        with torch.cuda.stream(self.s_geo):
            self.s_geo.wait_event(self.ev_lane[jc])
            if "side" not in skip:
                self.g_side[t].replay()
            self._set_valid[v1], self._lane_valid[jc][l] = self._lane_valid[jc][l], False
            launch = False
            if jl is not None:  # J new batches into the buffer whose lanes were all consumed J ticks ago
                got = 0
                if it is not None:
                    for k in range(J):
                        b = next(it, None)  # (a loader's host-to-device copies and feature assembly run here, on this queue)
                        if b is None:
                            break
                        assert b["pos"].shape[:2] == (self.B, self.N), "GraphPipeline: batches must keep the shape it was built for"
                        self._copy_batch(self.in_lane[jl][k], b)
                        got += 1
                self._lane_valid[jl] = [k < got for k in range(J)]
                launch = got > 0
            self.ev_rot.record(self.s_geo)
        if launch:
            with torch.cuda.stream(self.s_fps):
                self.s_fps.wait_event(self.ev_rot)
                if "fps" not in skip:
                    self.g_fps[jl].replay()
                self.ev_lane[jl].record(self.s_fps)
        with torch.cuda.stream(self.s_geo):
            if "geo" not in skip:
                self.g_geo[v1].replay()
            self.ev_geo.record(self.s_geo)
        return out, self._set_valid[v1] or any(self._lane_valid[0]) or any(self._lane_valid[1])

    def run(self, batches):
        """Generator: feeds `batches` (an iterable of device batch dicts) through the pipeline and yields one result per
        batch, in order: {'loss', 'logits', 'parts', 'target', 'data'} -- static tensors of the variant that just ran (read
        them on the current stream before taking the next-but-one result).  Ends when every batch has trained; an endless
        iterable makes an endless generator (bench.py)."""
        it = iter(batches)
        # start at a launch tick with empty buffers
        from . import schedule
        self.tick = 0
        self._set_valid = [False] * schedule.SETS
        self._lane_valid = [[False] * self.lanes, [False] * self.lanes]
        busy = True
        while busy:
            out, busy = self._tick(it)
            if out is not None:
                yield out

    # -- measurements (bench.py) ----------------------------------------------------------------------------------------
    def parts_alone(self, reps=5):
        """ms of each pipeline part replayed back to back on its own stream with nothing else on the chip: what the overlap
        has to hide.  (Leaves stale batches in the buffers: call after the timed run.)"""
        import time

        def alone(fn, stream):
            torch.cuda.synchronize()
            t = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t) / reps * 1e3, 3)
        return {"features_ms": alone(self.g_feat[0].replay, self.main),
                "update_ms": (0.0 if self.update_in_feature_graph else
                              alone(self.g_update.replay if self.g_update is not None else self._update, self.main)),
                "fps_all_levels_joint_launch_ms": alone(self.g_fps[0].replay, self.s_fps),
                "neighbourhood_geometry_ms": alone(self.g_geo[0].replay, self.s_geo),
                "hand_down_ms": alone(self.g_side[0].replay, self.s_geo)}

    def serial_ms(self, reps=5):
        """the same step with nothing overlapped: every part replayed on its stream with a host wait after each (a joint
        FPS launch once per J steps)"""
        import time
        J = self.lanes
        t = 0.0
        for r in range(-1, reps):
            if r == 0:
                torch.cuda.synchronize()
                t = time.perf_counter()
            for fn, st in ((self.g_side[r % len(self.g_side)].replay, self.s_geo),
                           (self.g_fps[(r // J) % 2].replay if r % J == 0 else (lambda: None), self.s_fps),
                           (self.g_geo[r % len(self.g_geo)].replay, self.s_geo), (self.g_feat[r % len(self.g_feat)].replay, self.main)):
                with torch.cuda.stream(st):
                    fn()
                st.synchronize()
            if self.flatg is not None and not self.collectives_captured:
                self.flatg.allreduce()
            if not self.update_in_feature_graph:
                (self.g_update.replay if self.g_update is not None else self._update)()
            self.main.synchronize()
        return round((time.perf_counter() - t) / reps * 1e3, 3)

    def describe(self):
        J = self.lanes
        seg = getattr(self.g_feat[0], "segments", 1)
        return {"launch": "hipGraph replay", "batches_per_joint_fps_launch": J, "look_ahead_batches": [J + 1, 2 * J],
                "geometry_queue_cus": self.geometry_cus or "all", "feature_graph_segments": seg,
                "collectives_per_step": (self.collectives_in_graph if self.collectives_captured else
                                         getattr(self.g_feat[0], "collectives", 0) + (1 if self.flatg is not None else 0)),
                "collectives": ("recorded into the feature graph (RCCL)" if self.collectives_captured else
                                "eager, between graph segments" if (self.sync_bn or self.flatg is not None) else "none"),
                "update": ("captured in the feature graph" if self.update_in_feature_graph else
                           "captured" if self.g_update is not None else "eager"),
                "pipeline": (f"3 queues: sampling (all FPS levels of {J} future batches as one launch every {J} steps) | neighbourhood + "
                             "loss geometry of the next batch (CU-masked) | features of this batch; geometry handed over without "
                             "copies (three captured variants each; the training stream carries no event another queue waits for)")}
