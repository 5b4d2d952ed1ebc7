"""Where the main stream's 0.3 ms per step of bubbles come from: the feature graph replayed back to back,
then with the pipeline's event traffic around it, then with (empty) side-queue launches per tick."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from amcontrast3d_amd import _lib, synthetic
from amcontrast3d_amd.pipeline import GraphPipeline
_lib.load()
cfg, model, criterion, aargs, opt = bench.build("S", dev, 1, False)
pool = [{k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(8, 24000, first_id=j * 8).items()} for j in range(2)]


def step_loss(data):
    logits, stage = model(data)
    return logits, criterion(logits, data["y"], stage, 13, None, aargs), ()


main = torch.cuda.Stream()
torch.cuda.set_stream(main)
pipe = GraphPipeline(model, step_loss, criterion.contrast_head, opt, pool[0], 13, None, aargs, lanes=4, max_grad_norm=10, keep_state=False)
run = pipe.run(itertools.cycle(pool))
for _ in range(20):
    next(run)
torch.cuda.synchronize()


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


g = pipe.g_feat[0]
print("feature graph back to back            %.3f ms" % timeit(g.replay))
ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()


def with_events():
    main.wait_event(ev_b)
    ev_a.record(main)
    g.replay()
    with torch.cuda.stream(pipe.s_geo):
        pipe.s_geo.wait_event(ev_a)
        ev_b.record(pipe.s_geo)


print("+ wait / record through the geometry queue %.3f ms" % timeit(with_events))


def with_side():
    main.wait_event(ev_b)
    ev_a.record(main)
    g.replay()
    with torch.cuda.stream(pipe.s_geo):
        pipe.s_geo.wait_event(ev_a)
        pipe.g_side[0].replay()
        ev_b.record(pipe.s_geo)


print("+ the hand-down graph on it             %.3f ms" % timeit(with_side))
alt = [pipe.g_feat[0], pipe.g_feat[1]]
it = itertools.cycle(alt)
print("two feature variants alternating        %.3f ms" % timeit(lambda: next(it).replay()))

# which of the two stream-level event operations costs, and does the kind of the other stream matter?
plain = torch.cuda.Stream()


def only_record():
    ev_a.record(main)
    g.replay()


def only_wait_completed():
    main.wait_event(ev_b)  # long complete
    g.replay()


def pair_with(stream):
    def f():
        main.wait_event(ev_b)
        ev_a.record(main)
        g.replay()
        with torch.cuda.stream(stream):
            stream.wait_event(ev_a)
            ev_b.record(stream)
    return f


print("record only                             %.3f ms" % timeit(only_record))
print("wait (on a completed event) only        %.3f ms" % timeit(only_wait_completed))
print("pair through a pooled torch stream      %.3f ms" % timeit(pair_with(plain)))
print("pair through the sampling queue         %.3f ms" % timeit(pair_with(pipe.s_fps)))
ev_c = torch.cuda.Event()


def record_after():
    # the hand-shake moved to the END of the graph: record after it, the other queue answers while the next graph is being launched
    main.wait_event(ev_b)
    g.replay()
    ev_a.record(main)
    with torch.cuda.stream(pipe.s_geo):
        pipe.s_geo.wait_event(ev_a)
        ev_b.record(pipe.s_geo)


print("record AFTER the graph (wait one step later) %.3f ms" % timeit(record_after))
try:
    ext_a, ext_b = torch.cuda.Event(external=True), torch.cuda.Event(external=True)
    print("external events are available")
except TypeError as e:
    print("no external events:", e)

# the same hand-shake with HIP events created with other release scopes (hip_runtime_api.h: hipEventDisableTiming 0x2,
# hipEventDisableSystemFence 0x20000000, hipEventReleaseToDevice 0x40000000, hipEventReleaseToSystem 0x80000000)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]


def raw_pair(flags):
    ea, eb = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(ea), flags) == 0 and hip.hipEventCreateWithFlags(ctypes.byref(eb), flags) == 0
    sm, sg = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(pipe.s_geo.cuda_stream)
    assert hip.hipEventRecord(eb, sg) == 0

    def f():
        assert hip.hipStreamWaitEvent(sm, eb, 0) == 0
        assert hip.hipEventRecord(ea, sm) == 0
        g.replay()
        assert hip.hipStreamWaitEvent(sg, ea, 0) == 0
        assert hip.hipEventRecord(eb, sg) == 0
    return f


for name, flags in (("disable-timing (torch's)", 0x2), ("+ release to device", 0x2 | 0x40000000), ("+ disable system fence", 0x2 | 0x20000000),
                    ("+ release to system", 0x2 | 0x80000000)):
    try:
        print("raw HIP events, %-26s %.3f ms" % (name, timeit(raw_pair(flags))))
    except AssertionError:
        print("raw HIP events, %-26s refused" % name)


# the host polls the other queue's event before it launches the next graph: the training stream then never carries a wait on
# an event that is incomplete at enqueue time (ROCclr drops waits on complete events)
def host_polled():
    while not ev_b.query():
        pass
    ev_a.record(main)
    g.replay()
    with torch.cuda.stream(pipe.s_geo):
        pipe.s_geo.wait_event(ev_a)
        ev_b.record(pipe.s_geo)


print("host polls ev_b, no stream wait         %.3f ms" % timeit(host_polled))


# one direction only: the training stream waits for the other queue's event, the other queue waits for nothing of the training
# stream (what a third result set + a host-side check of an old training-stream event would leave)
def one_way():
    main.wait_event(ev_b)
    g.replay()
    with torch.cuda.stream(pipe.s_geo):
        pipe.g_side[0].replay()
        ev_b.record(pipe.s_geo)


print("one way: main waits, the other queue never waits for main %.3f ms" % timeit(one_way))
ev_old = [torch.cuda.Event() for _ in range(3)]
cnt = [0]


def one_way_host_check():
    i = cnt[0]
    cnt[0] += 1
    ev_old[(i + 1) % 3].synchronize()  # the replay of two iterations ago has finished (host-side: no device dependency)
    main.wait_event(ev_b)
    g.replay()
    ev_old[i % 3].record(main)
    with torch.cuda.stream(pipe.s_geo):
        pipe.g_side[0].replay()
        ev_b.record(pipe.s_geo)


print("  + host-side check of the replay two iterations back      %.3f ms" % timeit(one_way_host_check))
