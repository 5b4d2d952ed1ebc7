"""Index arithmetic of the pipelined training loop (bench.py): which buffers a step consumes, reloads and launches.

The loop keeps four batches in flight:
    first-level FPS of batches t+3..   ->  FPS levels 2-4 of batch t+2  ->  neighbourhood / loss geometry of batch t+1  ->  features of t
and hands their inputs and results down one stage per step.  Two layouts of the first stage:
    lanes   `lanes` single-batch buffers, one launched per step (lane = step % lanes), consumed `lanes` steps later
    joint   two J-batch buffers (J = lanes), launched alternately every J steps; lane l of a launch is consumed J + l steps
            later -- the sampling queue then needs one launch every J steps
Kept apart from bench.py so that the hand-down can be replayed with batch ids instead of tensors (tests/test_host_logic.py)."""
import math


def period(lanes, npool, pingpong, joint):
    """Number of distinct step shapes (graphs are captured per shape): the lane and the pool index repeat after lcm(lanes,
    npool) steps, the ping-pong result sets after 2, the joint launches after 2 * lanes."""
    p = lanes * npool // math.gcd(lanes, npool)
    if pingpong and p % 2:
        p *= 2
    if joint:
        p = p * (2 * lanes) // math.gcd(p, 2 * lanes)
    return p


def side_step(s, lanes, joint, npool):
    """Step s (mod period) on the side queues -> dict
         consume  index into the first-level buffers whose result and inputs move down now: (buffer, lane) or (lane,)
         wait     the first-level launch (event index) that must have finished before that
         load     [(buffer index, pool index)]: inputs (re)loaded for the launch below
         launch   first-level launch started after the hand-down, or None
    joint: J = lanes batches per launch, one launch every J steps into buffer (s // J) % 2; lane l of a launch is consumed
    J + l steps after it was started (so a launch has J steps to finish)."""
    lane = s % lanes
    if joint:
        J = lanes
        assert J >= 2
        jc, jl = 1 - (s // J) % 2, (s // J) % 2
        first = lane == 0
        return {"consume": (jc, lane), "wait": jc, "launch": jl if first else None,
                "load": [((jl, t), (s + 2 * J + 1 + t) % npool) for t in range(J)] if first else []}
    return {"consume": (lane,), "wait": lane, "launch": lane, "load": [((lane,), (s + lanes + 3) % npool)]}


def variants(s, pingpong):
    """(result set the feature half of step s reads, result set stream B fills during step s)"""
    return (s % 2, (s + 1) % 2) if pingpong else (0, 0)
