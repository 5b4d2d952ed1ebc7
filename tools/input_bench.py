"""Throughput of the device input pipeline (amcontrast3d_amd/input_pipeline.py: voxelize + crop_pc of dataset/data_util.py:127-174)
on a raw synthetic room, next to the numpy oracle on the host:  python tools/input_bench.py [points=1200000] -> one JSON line.
A 'step' is what S3DIS.__getitem__ (dataset/s3dis/s3dis.py:122-144) does for one cloud: voxelize at 0.04 m, crop the 24000
nearest voxel representatives of a random centre, shift to the min corner."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amcontrast3d_amd import input_pipeline, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
dev = torch.device("cuda:0")
raw = synthetic.make_batch(1, n, first_id=77, voxel_size=0.005)
coord = torch.from_numpy(raw["pos"][0]).to(dev)
feat = torch.from_numpy(np.ascontiguousarray(raw["x"][0, :3].T)).to(dev)
label = torch.from_numpy(raw["y"][0]).to(dev)
gen = torch.Generator(device=dev).manual_seed(0)


def one():
    return input_pipeline.crop_pc(coord, feat, label, split="train", voxel_size=0.04, voxel_max=24000, generator=gen)


for _ in range(3):
    out = one()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    out = one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
line = {"metric": "input pipeline raw points/sec (voxelize 0.04 m + nearest-24000 crop, one cloud per call)", "value": round(n / dt, 1),
        "unit": "points/s", "ms_per_cloud": round(dt * 1e3, 3), "raw_points": n, "kept_points": int(out[0].shape[0])}
from oracle import input_ref  # (tools/: a measuring script, like bench.py's cpu_baseline leg)
c, l = coord.cpu().numpy(), label.cpu().numpy()
t0 = time.perf_counter()
c0 = c - c.min(0)
rng = np.random.default_rng(0)
key = input_ref.voxelize(c0, 0.04, mode=1)
uniq = input_ref.voxelize(c0, 0.04, mode=0, rnd=rng.integers(0, int(key[2].max()), key[2].size))
cv = c0[uniq]
input_ref.crop_nearest(cv, int(rng.integers(len(cv))), 24000)
line["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 1), "unit": "points/s", "kind": "port", "cores": 1,
                        "sample": "one cloud: oracle/input_ref.py voxelize (both modes, as crop_pc needs the counts) + crop_nearest, numpy"}
print(json.dumps(line))
