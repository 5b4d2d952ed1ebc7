import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
import amcontrast3d_amd; amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
def easy(d):
    c = EasyConfig(); c.update(d); return c
mode = sys.argv[1]
B, N = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model_from_cfg(easy(configs.model_cfg("S", dropout=0.5))).to(dev).train()
crit = build_criterion_from_cfg(easy(configs.criterion_cfg())).to(dev)
aa = easy(configs.ambiguity_args("s3dis"))
nb = synthetic.make_batch(B, N)
data = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
params = list(model.parameters())
opt = torch.optim.AdamW(params, lr=0.01, capturable=True)
out = {}
def fwd_bwd():
    logits, stage = model(data)
    if mode == "ce":
        out["loss"] = F.cross_entropy(logits.transpose(1, 2).reshape(-1, 13), data["y"].flatten())
    elif mode == "knn":
        from openpoints.cpp.pointops.functions import pointops
        st = stage["up"][0]
        idx, d = pointops.knnquery(24, st["p_out"], st["p_out"], st["offset"], st["offset"])
        out["loss"] = F.cross_entropy(logits.transpose(1, 2).reshape(-1, 13), data["y"].flatten()) + 0 * d.sum()
    else:
        out["loss"] = crit(logits, data["y"], stage, 13, None, aa)
    out["loss"].backward()
def update():
    torch.nn.utils.clip_grad_norm_(params, 10, norm_type=2)
    opt.step()
def eager():
    opt.zero_grad(set_to_none=True); fwd_bwd(); update()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): eager()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print(mode, "eager loss", float(out["loss"].detach()), flush=True)
opt.zero_grad(set_to_none=True)
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    fwd_bwd()
print("captured fwd_bwd", flush=True)
if "noupd" not in sys.argv:
    with torch.cuda.graph(g2, pool=g1.pool()):
        update()
    print("captured update", flush=True)
for i in range(3):
    g1.replay()
    torch.cuda.synchronize(); print("replay fb", i, float(out["loss"].detach()), flush=True)
    if "noupd" not in sys.argv:
        g2.replay(); torch.cuda.synchronize(); print("replay upd", i, flush=True)
print("OK", mode)
