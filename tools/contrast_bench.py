"""contrast_stage forward / backward on the loss stages of one synthetic S3DIS-like batch (8 x 24000), per stage, HIP-event
times; AMC3D_LIB selects a diagnostic build of the library (see scratch/contrast_diag.sh)"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, geometry, synthetic, ops
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0.5)); model = build_model_from_cfg(c).to(dev).train()
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(8, 24000).items()}
plan = geometry.precompute(model, crit.contrast_head, data, 13, None, aa)
torch.manual_seed(0)
for i, (g, C) in enumerate(zip(plan["loss"], (32, 64, 128, 256))):
    m = g["neighbor_idx"].shape[0]
    f = torch.randn(m, C, device=dev, requires_grad=True)
    sel = int(g["anchors"][0])
    def fwd():
        return ops.contrast_stage(f, g["neighbor_idx"], g["posmask"], g["ambiguity"], aa.mu, aa.nu, aa.temperature, g["anchors"], g.get("rev"), g.get("mutual"))
    for _ in range(3):
        fwd().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        f.grad = None
        e[0].record(); l = fwd(); e[1].record(); l.backward(); e[2].record(); torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]) / 10; tb += e[1].elapsed_time(e[2]) / 10
    mb = sel * 24 * C * 4 / 1e6
    print(f"stage {i}: m={m} C={C} selected={sel} ({100*sel/m:.1f} %)  fwd {tf*1e3:.0f} us  bwd {tb*1e3:.0f} us  rows {mb:.0f} MB -> bwd atomics at {mb/tb/1e3:.2f} TB/s")
