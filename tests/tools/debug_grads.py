import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from oracle import model_ref
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
def easy(d):
    c = EasyConfig(); c.update(d); return c
torch.manual_seed(0)
cfg = configs.model_cfg("S", dropout=0)
model = build_model_from_cfg(easy(cfg)).to(dev).train()
crit = build_criterion_from_cfg(easy(configs.criterion_cfg())).to(dev)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
nb = synthetic.make_batch(3, 3000, first_id=900)
cpu = {k: torch.from_numpy(v) for k, v in nb.items()}
gpu = {k: v.to(dev) for k, v in cpu.items()}
aa = configs.ambiguity_args("s3dis")
for mode, (w1, w2) in {"ce_only": (1.0, 0.0), "contrast_only": (0.0, 1.0), "both": (0.1, 0.9)}.items():
    a2 = dict(aa); a2["w1"], a2["w2"] = w1, w2
    want = model_ref.train_step(sd, cfg, cpu, cpu["y"], 13, None, a2)
    model.zero_grad()
    logits, stage = model(gpu)
    loss = crit(logits, gpu["y"], stage, 13, None, easy(a2))
    loss.backward()
    print(mode, "loss", float(loss), float(want["loss"]), "logit err", float((logits.detach().cpu() - want["logits"]).abs().max()))
    errs = []
    for k, p in model.named_parameters():
        ref = want["grads"][k]
        errs.append((float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12)), k, float(ref.norm())))
    errs.sort(reverse=True)
    for e in errs[:6]:
        print("   ", e)
    for i in range(4):
        fo = stage["up"][i]["f_out"].detach().cpu(); fr = want["stage"]["up"][i]["f_out"].detach()
        print("    f_out", i, float((fo - fr).abs().max()), float(fr.abs().max()))
