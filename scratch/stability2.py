"""train_one_epoch over many epochs with device-resident batches: ms/step and allocator growth"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic, train
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0.5)); model = build_model_from_cfg(c).to(dev)
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
cfg = EasyConfig(); cfg.update({"num_classes": 13, "ignore_index": None, "ambiguity_args": configs.ambiguity_args("s3dis"), "feature_keys": "x,heights",
                                "use_amp": False, "step_per_update": 1, "grad_norm_clip": 10, "sched_on_epoch": True})
opt = torch.optim.AdamW(model.parameters(), lr=0.01, fused=True)
pinned = []
for k in range(4):
    nb = synthetic.make_batch(8, 24000, first_id=8 * k)
    pinned.append({"pos": torch.from_numpy(nb["pos"]).pin_memory(), "y": torch.from_numpy(nb["y"]).pin_memory(),
                   "x": torch.from_numpy(np.ascontiguousarray(nb["x"][:, :3].transpose(0, 2, 1))).pin_memory(),
                   "heights": torch.from_numpy(np.ascontiguousarray(nb["x"][:, 3:4].transpose(0, 2, 1))).pin_memory()})
def loader(n):
    for k in range(n):
        yield dict(pinned[k % 4])
for ep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    torch.cuda.reset_peak_memory_stats(); t = time.perf_counter()
    out = train.train_one_epoch(model, loader(40), crit, opt, None, None, ep, cfg, prefetch_depth=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    torch.cuda.synchronize()
    print(f"epoch {ep}: loss {out[0]:.3f} {(time.perf_counter()-t)/40*1e3:.1f} ms/step peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
