"""which torch ops produce the copy / elementwise glue of one train step (torch.profiler, shapes recorded)"""
import sys, torch
sys.path.insert(0, ".")
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
def step():
    for p in model.parameters(): p.grad = None
    plan = geometry.precompute(model, crit.contrast_head, data, 13, None, aa)
    d = dict(data, _geometry=plan)
    logits, stage = model(d); loss = crit(logits, data["y"], stage, 13, None, aa); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
data2 = dict(data)
plan = geometry.precompute(model, crit.contrast_head, data, 13, None, aa)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for p in model.parameters(): p.grad = None
    d = dict(data, _geometry=plan)
    logits, stage = model(d); loss = crit(logits, data["y"], stage, 13, None, aa); loss.backward()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::add_", "aten::add", "aten::zeros", "aten::zero_", "aten::fill_",
                 "aten::gather", "aten::scatter_add_", "aten::index_select", "aten::sum", "aten::mul", "aten::transpose", "aten::max", "aten::relu", "aten::relu_",
                 "aten::threshold_backward", "aten::native_dropout", "aten::native_dropout_backward", "aten::sqrt", "aten::div", "aten::mean"):
        rows.append((e.device_time_total, e.key, e.count, str(e.input_shapes)[:110]))
for t, k, n, sh in sorted(rows, reverse=True)[:40]:
    print(f"{t:8.0f} us  {k:28s} x{n:3d}  {sh}")
