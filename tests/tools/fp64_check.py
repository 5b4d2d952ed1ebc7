import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import amcontrast3d_amd; amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from oracle import model_ref
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
def easy(d):
    c = EasyConfig(); c.update(d); return c
torch.manual_seed(0)
cfg = configs.model_cfg("S", dropout=0)
model = build_model_from_cfg(easy(cfg))
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
nb = synthetic.make_batch(3, 3000, first_id=900)
cpu = {k: torch.from_numpy(v) for k, v in nb.items()}
aa = dict(configs.ambiguity_args("s3dis")); aa["w1"], aa["w2"], aa["stages_num"] = 1.0, 0.0, 0
r32 = model_ref.train_step(sd, cfg, cpu, cpu["y"], 13, None, aa)
# fp64: same neighbour structure (indices come from fp32 xyz), features/weights in double
import oracle.pointops_ref as K
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
orig = (K.ball_query, K.furthest_point_sample, K.three_nn, K.knnquery)
K.ball_query = lambda r, n, x, q: orig[0](r, n, x.float().contiguous(), q.float().contiguous())
K.furthest_point_sample = lambda x, n: orig[1](x.float().contiguous(), n)
K.three_nn = lambda u, k: tuple(t.double() if t.dtype.is_floating_point else t for t in orig[2](u.float().contiguous(), k.float().contiguous()))
K.knnquery = lambda ns, x, q, o, qo: orig[3](ns, x.float().contiguous(), q.float().contiguous(), o, qo)
data64 = {"pos": cpu["pos"].double(), "x": cpu["x"].double()}
r64 = model_ref.train_step(sd64, cfg, data64, cpu["y"], 13, None, aa)
errs = sorted(((float((r32["grads"][k].double() - r64["grads"][k]).norm() / r64["grads"][k].norm()), k) for k in r64["grads"]), reverse=True)
print("fp32-CPU vs fp64 grads (CE only): worst", errs[:5])
print("logits err", float((r32["logits"].double() - r64["logits"]).abs().max()))
torch.save({k: v for k, v in r64["grads"].items()}, "scratch/grads64.pt")
