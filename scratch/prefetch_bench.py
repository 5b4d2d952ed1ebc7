"""eager training loop (no hipGraph) at the benchmark shape: plain vs GeometryPrefetcher"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from amcontrast3d_amd.pipeline import GeometryPrefetcher
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device('cuda:0')
torch.manual_seed(0)
c = EasyConfig(); c.update(configs.model_cfg("S"))
cc = EasyConfig(); cc.update(configs.criterion_cfg())
aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
model = build_model_from_cfg(c).to(dev).train(); criterion = build_criterion_from_cfg(cc).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, fused=True)
nb = synthetic.make_batch(8, 24000)
base = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
def batches(n):
    for _ in range(n): yield dict(base)
def loop(src, n):
    t0 = None; k = 0
    for data in src:
        if k == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        logits, stage = model(data)
        loss = criterion(logits, data['y'], stage, 13, None, aa)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); k += 1
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (k - 3) * 1e3
print(f"plain eager loop      : {loop(batches(13), 13):7.2f} ms/step")
for d in (1, 2, 3):
    print(f"prefetched (depth {d})  : {loop(GeometryPrefetcher(batches(13), model, criterion.contrast_head, 13, None, aa, depth=d), 13):7.2f} ms/step")
