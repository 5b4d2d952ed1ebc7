"""Index arithmetic of the pipelined training loop (pipeline.GraphPipeline): which buffers a tick trains on, fills, consumes
and launches.  Kept apart from the streams and graphs so that the hand-down can be replayed with batch ids instead of tensors
(tests/test_host_logic.py).

Buffers:
    joint[j], j = 0, 1      two J-batch input buffers of the sampling queue with their FPS results (all levels); buffer j is
                            loaded and launched every 2J ticks, J ticks apart from the other one
    set[v],  v = 0, 1, 2    one batch + its FPS picks + (written by geometry variant v) its neighbourhood / loss geometry;
                            read by feature variant v one tick after geometry variant v filled it.  THREE sets, not two: the set
                            filled at tick t was last read at tick t - 2, which the HOST can check (an event it merely queries) --
                            the geometry queue then never waits for an event of the training stream, and such an event costs the
                            training stream 0.2 ms per step (tools/bubble_probe.py)
A tick t (mod lcm(2J, 3)):
    train    set[t % 3]                                   (main stream)
    consume  lane (jc, l) -> set[(t + 1) % 3]             (geometry queue; jc = the buffer launched J..2J-1 ticks ago, l = t % J)
    launch   buffer jl, after loading J new batches into it, when l == 0     (sampling queue; has J ticks to finish)
so a batch loaded at a launch tick t0 into lane l is consumed at t0 + J + l and trains at t0 + J + l + 1: batches leave in the
order they entered, 1 .. J ticks after their sampling finished at the latest."""


def tick_plan(t, lanes):
    """-> {'train': set index, 'fill': set index, 'consume': (joint buffer, lane), 'launch': joint buffer or None}"""
    J = lanes
    assert J >= 2
    t %= period(J)
    jl = (t // J) % 2
    return {"train": t % SETS, "fill": (t + 1) % SETS, "consume": (1 - jl, t % J), "launch": jl if t % J == 0 else None}


SETS = 3


def period(lanes):
    """distinct tick shapes (hand-down graphs are captured per shape): lcm(2J, 3)"""
    import math
    return 2 * lanes * SETS // math.gcd(2 * lanes, SETS)
