"""AMContrast3D++ loop and whole-room testing over many repetitions: ms/step and allocator growth"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, evaluate, synthetic, train
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
c = EasyConfig(); c.update(configs.model_cfg_mm("S", dropout=0.5)); model = build_model_from_cfg(c).to(dev)
cc = EasyConfig(); cc.update(configs.criterion_cfg_mm()); crit = build_criterion_from_cfg(cc).to(dev)
cfg = EasyConfig(); cfg.update({"num_classes": 13, "ignore_index": None, "ambiguity_args": configs.ambiguity_args_mm("s3dis"), "feature_keys": "x,heights",
                                "use_amp": False, "step_per_update": 1, "grad_norm_clip": 10, "sched_on_epoch": True})
opt = torch.optim.AdamW(model.parameters(), lr=0.01, fused=True)
pinned = []
for k in range(4):
    nb = synthetic.make_batch(8, 24000, first_id=8 * k)
    pinned.append({"pos": torch.from_numpy(nb["pos"]).pin_memory(), "y": torch.from_numpy(nb["y"]).pin_memory(),
                   "x": torch.from_numpy(np.ascontiguousarray(nb["x"][:, :3].transpose(0, 2, 1))).pin_memory(),
                   "heights": torch.from_numpy(np.ascontiguousarray(nb["x"][:, 3:4].transpose(0, 2, 1))).pin_memory()})
for ep in range(5):
    t = time.perf_counter()
    out = train.train_one_epoch_mm(model, (dict(pinned[k % 4]) for k in range(30)), crit, opt, None, None, ep, cfg)
    torch.cuda.synchronize()
    print(f"mm epoch {ep}: loss {out[0]:.3f} rate {out[5]:.1f}% {(time.perf_counter()-t)/30*1e3:.1f} ms/step reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
room = synthetic.make_batch(1, 300000, first_id=900, voxel_size=0.02)
coord = room["pos"][0] - room["pos"][0].min(0); feat = room["x"][0, :3].T.copy()
label = torch.from_numpy(room["y"][0].astype(np.int64)).to(dev)
parts = evaluate.voxel_parts(coord, 0.04)
c2 = EasyConfig(); c2.update(configs.model_cfg("S", dropout=0.5)); m2 = build_model_from_cfg(c2).to(dev)
for rep in range(4):
    t = time.perf_counter()
    for _ in range(10):
        evaluate.test_cloud_boundary_inner(m2, coord, feat, label, parts, 13, None, 24)
    torch.cuda.synchronize()
    print(f"rooms x10: {(time.perf_counter()-t)/10*1e3:.1f} ms/room reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
