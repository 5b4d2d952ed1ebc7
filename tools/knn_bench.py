import sys, os
sys.path.insert(0, os.getcwd())
import torch
from amcontrast3d_amd import ops, synthetic
dev = torch.device("cuda:0")
B, N = 8, 24000
nb = synthetic.make_batch(B, N)
pos = torch.from_numpy(nb["pos"]).to(dev)
def flat(p): return p.reshape(-1, 3).contiguous()
def off(p): return torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
stages = [flat(pos)]
cur = pos
for m in (6000, 1500, 375):
    idx = ops.furthest_point_sample(cur, m).long()
    cur = torch.gather(cur, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    stages.append(flat(cur))
def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for i, p in enumerate(stages):
    o = off(p)
    print(f"self  stage {i} n={p.shape[0]:7d} k=24: {tm(lambda: ops.knnquery(24, p, p, o, o)):.3f} ms")
for i, kr in ((1, 4), (2, 16), (3, 64)):
    p = stages[i]
    print(f"label stage {i} m={p.shape[0]:7d} kr={kr}: {tm(lambda: ops.knnquery(kr, stages[0], p, off(stages[0]), off(p))):.3f} ms")
