import sys, os
sys.path.insert(0, os.getcwd())
import torch, ctypes
from amcontrast3d_amd import _lib, synthetic
lib = _lib.load()
dev = torch.device("cuda:0")
B, N, M = 8, int(sys.argv[1]), int(sys.argv[2])
xyz = torch.from_numpy(synthetic.make_batch(B, N)["pos"]).to(dev)
temp = torch.full((B, N), 1e10, device=dev)
out = torch.empty(B, M, dtype=torch.int32, device=dev)
work = torch.empty(B * N, dtype=torch.int32, device=dev)
st = lib.amc3d_furthest_point_sampling(B, N, M, xyz.data_ptr(), temp.data_ptr(), out.data_ptr(), work.data_ptr(), work.numel() * 4, None)
torch.cuda.synchronize()
t = temp[0].cpu()
nw = 8
print(f"N={N} M={M} cycles per iteration (s_memtime ticks) per wave: [sweep, lane+wave argmax+record, barrier wait, select]")
for w in range(nw):
    v = t[w * 8: w * 8 + 4] / (M - 1)
    print(w, [round(float(x), 1) for x in v], "sum", round(float(v.sum()), 1))
