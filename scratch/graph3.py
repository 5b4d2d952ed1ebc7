import sys, os
sys.path.insert(0, os.getcwd())
import torch
import amcontrast3d_amd; amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic, geometry
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
def easy(d):
    c = EasyConfig(); c.update(d); return c
dev = torch.device("cuda:0")
model = build_model_from_cfg(easy(configs.model_cfg("S", dropout=0))).to(dev).train()
crit = build_criterion_from_cfg(easy(configs.criterion_cfg())).to(dev)
aa = easy(configs.ambiguity_args("s3dis"))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(2, 4096).items()}
side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
def plan():
    return geometry.precompute(model, crit.contrast_head, data, 13, None, aa, aux_stream=side2, join=False)
p0 = plan(); torch.cuda.synchronize(); print("eager plan ok", flush=True)
g = torch.cuda.CUDAGraph()
keep = []
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur); side2.wait_stream(cur)
    with torch.cuda.stream(side):
        keep.append(plan())
    cur.wait_stream(side); cur.wait_stream(side2)
print("captured", flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", i, flush=True)
same = torch.equal(keep[0]["loss"][0]["ambiguity"], p0["loss"][0]["ambiguity"]) and torch.equal(keep[0]["encoder"][4][0]["idx"], p0["encoder"][4][0]["idx"])
print("OK same=", same)
