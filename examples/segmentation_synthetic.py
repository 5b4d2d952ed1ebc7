"""End-to-end run of the framework on synthetic S3DIS-shaped rooms: build PointNeXt + AMContrast3D from the reference's
config keys, train a few epochs with `amcontrast3d_amd.train.train_one_epoch` (the reference's loop, geometry
prefetched), validate with boundary / inner mIoU, test one whole room with sub-cloud voting.

    python examples/segmentation_synthetic.py [--epochs 3] [--batches 12] [--variant S] [--points 24000]

There are no datasets in this repository (no network): `amcontrast3d_amd.synthetic` generates rooms of planes and
boxes with spatially coherent labels, which a model can learn within a few dozen steps.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amcontrast3d_amd  # noqa: E402

amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, evaluate, synthetic, train  # noqa: E402
from openpoints.loss import build_criterion_from_cfg  # noqa: E402
from openpoints.models import build_model_from_cfg  # noqa: E402
from openpoints.utils import EasyConfig  # noqa: E402


PALETTE = np.random.default_rng(0).random((13, 3)).astype(np.float32)


def colour_by_class(nb, seed):
    """The benchmark generator draws colours at random (labels are then unlearnable from the features, which is
    irrelevant for timing); here colour = a class colour + noise, so that a few dozen steps show learning."""
    noise = np.random.default_rng(seed).random(nb["x"][:, :3].shape).astype(np.float32)
    nb["x"][:, :3] = 0.7 * PALETTE[nb["y"]].transpose(0, 2, 1) + 0.3 * noise
    return nb


def loader(first_id, n_batches, batch, points):
    """batches in the reference's collated layout: point-major 'x' (colour) and 'heights', 'y' (B,N)"""
    for k in range(n_batches):
        nb = colour_by_class(synthetic.make_batch(batch, points, first_id=first_id + k * batch), first_id + k)
        yield {"pos": torch.from_numpy(nb["pos"]), "y": torch.from_numpy(nb["y"]),
               "x": torch.from_numpy(np.ascontiguousarray(nb["x"][:, :3].transpose(0, 2, 1))),
               "heights": torch.from_numpy(np.ascontiguousarray(nb["x"][:, 3:4].transpose(0, 2, 1)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=24000)
    ap.add_argument("--variant", default="S")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    c = EasyConfig(); c.update(configs.model_cfg(args.variant, dropout=0.5))
    model = build_model_from_cfg(c).to(dev)
    cc = EasyConfig(); cc.update(configs.criterion_cfg())
    criterion = build_criterion_from_cfg(cc).to(dev)
    cfg = EasyConfig()
    cfg.update({"num_classes": 13, "ignore_index": None, "ambiguity_args": configs.ambiguity_args("s3dis"),
                "feature_keys": "x,heights", "use_amp": False, "step_per_update": 1, "grad_norm_clip": 10,
                "sched_on_epoch": True})
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-4)  # cfgs/s3dis/default.yaml:64-72
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.epochs)
    for epoch in range(1, args.epochs + 1):
        t0 = time.perf_counter()
        loss, miou, macc, oa, _, _ = train.train_one_epoch(model, loader(10000 * epoch, args.batches, args.batch, args.points),
                                                          criterion, opt, sched, None, epoch, cfg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sched.step()
        val = ({k: v.to(dev) for k, v in d.items()} for d in loader(900000, 4, 1, args.points))
        val = ({**d, "x": train.get_features_by_keys(d, cfg.feature_keys)} for d in val)
        v = evaluate.validate_boundary_inner(model, val, 13, None, cfg.ambiguity_args.nsample)
        print(f"epoch {epoch}: loss {loss:.3f} train mIoU {miou:.1f} OA {oa:.1f} | val mIoU {v[0]:.1f} boundary {v[5]:.1f} "
              f"inner {v[10]:.1f} | {args.batches * args.batch * args.points / dt / 1e6:.2f} M points/s incl. host data generation")
    room = colour_by_class(synthetic.make_batch(1, 200000, first_id=777, voxel_size=0.02), 777)
    coord = room["pos"][0] - room["pos"][0].min(0)
    label = torch.from_numpy(room["y"][0].astype(np.int64)).to(dev)
    parts = evaluate.voxel_parts(coord, 0.04)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = evaluate.test_cloud_boundary_inner(model, coord, room["x"][0, :3].T.copy(), label, parts, 13, None, 24)
    torch.cuda.synchronize()
    s = evaluate.summarize(r["cm"], r["cm_b"], r["cm_i"])
    print(f"whole room ({len(coord)} points, {len(parts)} sub-clouds, {time.perf_counter() - t0:.3f} s): mIoU {s[0]:.1f} OA {s[2]:.1f} "
          f"boundary mIoU {s[5]:.1f}")


if __name__ == "__main__":
    main()
