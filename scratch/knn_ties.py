import sys, os
sys.path.insert(0, os.getcwd())
import torch
from amcontrast3d_amd import ops, synthetic, _lib
dev = torch.device("cuda:0")
B, N = 8, 24000
nb = synthetic.make_batch(B, N)
pos = torch.from_numpy(nb["pos"]).to(dev)
p = pos.reshape(-1, 3).contiguous()
o = torch.tensor([p.shape[0]], dtype=torch.int32, device=dev)
lib = _lib.load()
n = p.shape[0]
for k in (24, 4):
    wb = int(lib.amc3d_knnquery_workspace_bytes(n, n, k, 1))
    work = torch.zeros(wb, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, k, dtype=torch.int32, device=dev); d2 = torch.empty(n, k, device=dev)
    st = lib.amc3d_knnquery(n, k, n, 1, p.data_ptr(), p.data_ptr(), o.data_ptr(), o.data_ptr(), idx.data_ptr(), d2.data_ptr(), work.data_ptr(), wb, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("k", k, "status", st, "fallback queries:", int(work[1024:1028].view(torch.int32)[0]), "of", n)
    gp = work[0:40].view(torch.float32)
    print("   h", float(gp[3]), "dims", work[24:40].view(torch.int32).tolist())
# how many exact duplicate points?
u = torch.unique(p, dim=0)
print("unique points", u.shape[0], "of", n)
