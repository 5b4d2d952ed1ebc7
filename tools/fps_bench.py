import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from amcontrast3d_amd import ops, synthetic
dev = torch.device("cuda:0")
B, N, M = 8, int(sys.argv[1]), int(sys.argv[2])
xyz = torch.from_numpy(synthetic.make_batch(B, N)["pos"]).to(dev)
for _ in range(2): ops.furthest_point_sample(xyz, M)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): idx = ops.furthest_point_sample(xyz, M)
e.record(); torch.cuda.synchronize()
print(f"dbg={os.environ.get('AMC3D_FPS_DEBUG','0')} N={N} M={M}: {s.elapsed_time(e)/5:.3f} ms/call  ({s.elapsed_time(e)/5/M*1e3:.3f} us/iter)  sum={int(idx.sum())}")
