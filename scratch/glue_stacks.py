"""where the ATen glue kernels of one train step come from: torch.profiler with python stacks, grouped by the innermost
frame inside this package (file:line) and the op name -- device time and launch count per group"""
import sys, os, collections, torch
sys.path.insert(0, os.getcwd())
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, geometry, synthetic
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0.5)); model = build_model_from_cfg(c).to(dev).train()
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(8, 24000).items()}
plan = geometry.precompute(model, crit.contrast_head, data, 13, None, aa)
def step():
    for p in model.parameters(): p.grad = None
    d = dict(data, _geometry=plan)
    logits, stage = model(d); loss = crit(logits, data["y"], stage, 13, None, aa); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
groups = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    t_dev = getattr(e, "self_device_time_total", 0) or 0
    if not e.name.startswith("aten::") or t_dev <= 0:  # aten ops that launched kernels themselves
        continue
    where = "autograd / torch"
    for fr in (e.stack or []):
        if "amcontrast3d_amd" in fr and "site-packages" not in fr:
            where = fr.split("amcontrast3d_amd/")[-1][:70]
            break
    g = groups[(where, e.name)]
    g[0] += t_dev; g[1] += 1
rows = sorted(groups.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in groups.values())
print(f"leaf aten ops with device time: {tot:.0f} us in {sum(v[1] for v in groups.values())} launches")
for (where, name), (t, n) in rows[:45]:
    print(f"{t:7.0f} us x{n:3d}  {name:28s} {where}")
