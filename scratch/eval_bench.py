"""Whole-room evaluation throughput (amcontrast3d_amd.evaluate.test_cloud_boundary_inner) on the MI355X, with the
per-sub-cloud forward and boundary-mask times (the CPU baseline is in `bench.py --eval`).
python scratch/eval_bench.py [room_points] [variant]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, evaluate, synthetic
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig

n_room = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
variant = sys.argv[2] if len(sys.argv) > 2 else "S"
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = EasyConfig(); c.update(configs.model_cfg(variant, dropout=0.5))
model = build_model_from_cfg(c).to(dev).eval()
t = time.perf_counter()
room = synthetic.make_batch(1, n_room, first_id=900, voxel_size=0.02)
coord = room["pos"][0] - room["pos"][0].min(0); feat = room["x"][0, :3].T.copy()
label = torch.from_numpy(room["y"][0].astype(np.int64)).to(dev)
parts = evaluate.voxel_parts(coord, 0.04)
print(f"room {n_room} points, {len(parts)} sub-clouds of {len(parts[0])} points (host prep {time.perf_counter()-t:.1f} s)", file=sys.stderr)
def run(miou_B_I):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = evaluate.test_cloud_boundary_inner(model, coord, feat, label, parts, 13, None, 24, miou_B_I=miou_B_I)
    torch.cuda.synchronize(); return time.perf_counter() - t, r
run(True)
dt_full = min(run(True)[0] for _ in range(3))
dt_plain = min(run(False)[0] for _ in range(3))
# model forward alone on one sub-cloud, inputs resident
cp = coord[parts[0]] - coord[parts[0]].min(0)
pos = torch.from_numpy(np.ascontiguousarray(cp, dtype=np.float32)).to(dev).unsqueeze(0)
x = torch.cat([torch.from_numpy(feat[parts[0]]).to(dev), pos[0, :, 2:3]], 1).t().contiguous().unsqueeze(0)
with torch.no_grad():
    for _ in range(2): model({"pos": pos, "x": x})
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): model({"pos": pos, "x": x})
    torch.cuda.synchronize(); fwd = (time.perf_counter() - t) / 5
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): evaluate.boundary_mask(pos[0], label[torch.from_numpy(parts[0]).to(dev)], 24, 13, None)
    torch.cuda.synchronize(); bm = (time.perf_counter() - t) / 5
out = {"room_points": n_room, "sub_clouds": len(parts), "points_per_sub_cloud": len(parts[0]), "variant": variant,
       "whole_room_s": round(dt_full, 4), "whole_room_without_boundary_split_s": round(dt_plain, 4),
       "sub_cloud_points_per_s": round(len(parts) * len(parts[0]) / dt_full), "forward_ms_per_sub_cloud": round(fwd * 1e3, 2),
       "boundary_mask_ms_per_sub_cloud": round(bm * 1e3, 2)}
print(json.dumps(out))
