"""where the wall time of evaluate.test_cloud_boundary_inner goes (300k-point room): input staging, model calls with
and without geometry prefetch, boundary masks, vote"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, evaluate, geometry, synthetic
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0.5))
model = build_model_from_cfg(c).to(dev).eval()
room = synthetic.make_batch(1, 300000, first_id=900, voxel_size=0.02)
coord = room["pos"][0] - room["pos"][0].min(0); feat = room["x"][0, :3].T.copy()
label = torch.from_numpy(room["y"][0].astype(np.int64)).to(dev)
parts = evaluate.voxel_parts(coord, 0.04)
def T(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
def stage():
    out = []
    for part in parts:
        cp = coord[part]; cp = cp - cp.min(0)
        pos = torch.from_numpy(np.ascontiguousarray(cp, dtype=np.float32)).to(dev).unsqueeze(0)
        x = torch.cat([torch.from_numpy(feat[part]).to(dev), pos[0, :, 2:3]], 1).t().contiguous().unsqueeze(0)
        out.append({"pos": pos, "x": x})
    return out
t_stage, inputs = T(stage)
stacks = [{k: torch.cat([d[k] for d in inputs[j:j + 8]], 0) for k in inputs[0]} for j in range(0, len(inputs), 8)]
with torch.no_grad():
    t_model, _ = T(lambda: [model(s)[0] for s in stacks])
    t_geo, plans = T(lambda: [geometry.precompute_sampling(model, s) for s in stacks])
    t_feat, _ = T(lambda: [model(dict(s, _geometry=p))[0] for s, p in zip(stacks, plans)])
    t_bm, _ = T(lambda: [evaluate.boundary_mask(d["pos"][0], label[torch.from_numpy(p).to(dev)], 24, 13, None) for d, p in zip(inputs, parts)])
for bs in (8, 64):
    t_all, _ = T(lambda: evaluate.test_cloud_boundary_inner(model, coord, feat, label, parts, 13, None, 24, batch=bs))
    print(f"test_cloud_boundary_inner batch={bs}: {t_all:.1f} ms")
print(f"stage inputs {t_stage:.1f} ms | model (5 stacks) {t_model:.1f} = geometry {t_geo:.1f} + features {t_feat:.1f} | boundary masks (39) {t_bm:.1f} ms")
