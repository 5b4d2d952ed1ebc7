"""torch-generated nodes of one train step's autograd graph (each is at least one ATen launch in backward), grouped by node
type and by the line of this package that created them in forward (anomaly mode keeps the forward traceback)"""
import sys, os, collections, torch
sys.path.insert(0, os.getcwd())
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, geometry, synthetic
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "S"
c = EasyConfig(); c.update(configs.model_cfg(variant, dropout=0.5)); model = build_model_from_cfg(c).to(dev).train()
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(8, 24000).items()}
plan = geometry.precompute(model, crit.contrast_head, data, 13, None, aa)
with torch.autograd.set_detect_anomaly(True, check_nan=False):
    d = dict(data, _geometry=plan)
    logits, stage = model(d); loss = crit(logits, data["y"], stage, 13, None, aa)
seen, todo = set(), [loss.grad_fn]
groups = collections.Counter()
fanout = collections.Counter()
while todo:
    n = todo.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    name = type(n).__name__
    for nxt, _ in n.next_functions:
        if nxt is not None:
            fanout[nxt] += 1
            todo.append(nxt)
    if name in ("AccumulateGrad",) or name.endswith("Backward") and not name[0].isupper():
        continue
    tb = n.metadata.get("traceback_", [])
    where = "?"
    for line in reversed(tb):
        if "amcontrast3d_amd" in line and "File" in line:
            where = line.strip().split("amcontrast3d_amd/")[-1].replace('", line ', ':').split(",")[0]
            break
    groups[(name, where)] += 1
for (name, where), k in sorted(groups.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{k:4d}  {name:34s} {where}")
print("--- nodes whose output gradient is summed from several consumers (one add per extra consumer):")
for n, k in sorted(fanout.items(), key=lambda kv: -kv[1]):
    if k > 1 and type(n).__name__ != "AccumulateGrad":
        tb = n.metadata.get("traceback_", [])
        where = "?"
        for line in reversed(tb):
            if "amcontrast3d_amd" in line and "File" in line:
                where = line.strip().split("amcontrast3d_amd/")[-1].replace('", line ', ':').split(",")[0]
                break
        print(f"{k:4d}  {type(n).__name__:34s} {where}")
