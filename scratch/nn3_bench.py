import sys, os
sys.path.insert(0, os.getcwd())
import torch
from amcontrast3d_amd import ops, synthetic
dev = torch.device("cuda:0")
pos = torch.from_numpy(synthetic.make_batch(8, 24000)["pos"]).to(dev)
cur = pos; lv = [pos]
for m in (6000, 1500):
    idx = ops.furthest_point_sample(cur, m).long()
    cur = torch.gather(cur, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous(); lv.append(cur)
def tm(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
print(f"three_nn 24000<-6000: {tm(lambda: ops.three_nn(lv[0], lv[1])):.3f} ms   6000<-1500: {tm(lambda: ops.three_nn(lv[1], lv[2])):.3f} ms   ball 6000 in 24000: {tm(lambda: ops.ball_query(0.1, 32, lv[0], lv[1])):.3f} ms")
