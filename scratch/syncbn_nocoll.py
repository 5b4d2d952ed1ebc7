"""bench.py --sync-bn on a one-rank group with the all-reduces stubbed out: separates the cost of the extra BN launches
from the cost of RCCL nodes inside the captured step (AMC3D_FORCE_SYNC_BN=1 python scratch/syncbn_nocoll.py ...)"""
import runpy, sys
import torch.distributed as dist
_real = dist.all_reduce
calls = [0]
def fake(t, *a, **k):
    calls[0] += 1
dist.all_reduce = fake
sys.argv = ["bench.py", "--sync-bn"] + sys.argv[1:]
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    print("stubbed all_reduce calls:", calls[0], file=sys.stderr)
