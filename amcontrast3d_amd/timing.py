"""HIP-event timing of the native operators, on the stream they are launched on.

bench.py switches this on for the timed region: every C-ABI launch made through
amcontrast3d_amd.ops is bracketed by two events recorded on torch's current stream (the
stream the kernel is enqueued on), tagged with the operator name and two byte counts:

    bytes   ALGORITHMIC bytes of the launch in the sense of SURVEY.md section 8(d): every input read once, every output
            written once -- the compulsory HBM traffic of the operator however it is implemented
    moved   what THIS implementation asks the memory system for: re-reads of a multi-pass kernel, gathered neighbour rows,
            rows added by float atomics.  Most of the excess is served by L2 / Infinity Cache; it is reported next to the
            algorithmic figure, never instead of it (round 2's bench line priced the loss backward with it: 0.38 claimed,
            0.021 by section 8(d))

Nothing synchronises until ``collect()``.
"""
import collections

import torch

_enabled = False
_records = []  # (name, start_event, end_event, algorithmic_bytes, flops, moved_bytes)
calls = None   # a collections.Counter while count_calls() is active: operator name -> launches (dispatch evidence)


class count_calls:
    """with timing.count_calls() as c: ... -> c[name] = number of C-ABI launches of that operator inside the block
    (no events, no synchronisation): the parity tests assert with it that a size took the kernels the bench times."""

    def __enter__(self):
        global calls
        self.prev = calls
        calls = collections.Counter()
        return calls

    def __exit__(self, *exc):
        global calls
        calls = self.prev
        return False


def enable(flag=True):
    global _enabled
    _enabled = bool(flag)
    if not flag:
        _records.clear()


def enabled():
    return _enabled


def note(name):
    """count a dispatch that is not a C-ABI launch of its own (e.g. a library GEMM) while count_calls() is active"""
    if calls is not None:
        calls[name] += 1


class span:
    """with timing.span('knnquery', bytes): launch(...)"""
    __slots__ = ("name", "nbytes", "flops", "moved", "start")

    def __init__(self, name, nbytes=0, flops=0.0, moved=None):
        self.name, self.nbytes, self.flops, self.start = name, nbytes, flops, None
        self.moved = nbytes if moved is None else moved
        if calls is not None:
            calls[name] += 1

    def __enter__(self):
        if _enabled:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _records.append((self.name, self.start, end, self.nbytes, self.flops, self.moved))
        return False


def collect():
    """-> {name: {'launches', 'total_ms', 'avg_ms', 'bytes', 'moved', 'flops', 'bytes_per_launch'}}; call after a device sync."""
    out = collections.OrderedDict()
    for name, s, e, nbytes, flops, moved in _records:
        d = out.setdefault(name, {"launches": 0, "total_ms": 0.0, "bytes": 0, "flops": 0.0, "moved": 0})
        d["launches"] += 1
        d["total_ms"] += s.elapsed_time(e)
        d["bytes"] += nbytes
        d["flops"] += flops
        d["moved"] += moved
    for d in out.values():
        d["avg_ms"] = d["total_ms"] / d["launches"]
        d["bytes_per_launch"] = d["bytes"] / d["launches"]
    _records.clear()
    return out
