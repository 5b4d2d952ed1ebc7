"""Seeded synthetic S3DIS-/ScanNet-shaped scenes.

There is no dataset in the build image or on the GPU box, so benchmarks and
parity tests run on generated rooms with the statistics the reference's loader
produces (openpoints/dataset/data_util.py:127-174, dataset/s3dis/s3dis.py:122-144):
points on surfaces, one point per ``voxel_size`` voxel, the ``N`` points nearest
to a random centre (``crop_pc``), shuffled, shifted to min-corner 0, labels that
are spatially coherent (one class per surface), colours in [0,1) and a height
channel.  Everything is numpy + a fixed seed so a sample is reproducible
bit-for-bit on every machine.
"""
import numpy as np


def _room_surfaces(rng):
    """Six walls of a 6 x 5 x 3 m shell plus eight boxes; returns a list of
    (origin, edge_u, edge_v, class_id) rectangles."""
    rects = []
    L = np.array([6.0, 5.0, 3.0]) * rng.uniform(0.85, 1.15, size=3)
    cid = 0
    for ax in range(3):
        u, v = [a for a in range(3) if a != ax]
        for side in (0.0, L[ax]):
            o = np.zeros(3); o[ax] = side
            eu = np.zeros(3); eu[u] = L[u]
            ev = np.zeros(3); ev[v] = L[v]
            rects.append((o, eu, ev, cid)); cid += 1
    for _ in range(8):
        size = rng.uniform(0.3, 1.4, size=3)
        lo = rng.uniform(0.0, 1.0, size=3) * (L - size)
        lo[2] = 0.0 if rng.uniform() < 0.7 else lo[2]
        for ax in range(3):
            u, v = [a for a in range(3) if a != ax]
            for side in (lo[ax], lo[ax] + size[ax]):
                o = lo.copy(); o[ax] = side
                eu = np.zeros(3); eu[u] = size[u]
                ev = np.zeros(3); ev[v] = size[v]
                rects.append((o, eu, ev, cid))
        cid += 1
    return rects


def make_scene(sample_id, n_points, voxel_size=0.04, num_classes=13, ignore_frac=0.0,
               ignore_index=-100, duplicates=False, seed=1234):
    """One cloud: dict(pos (N,3) f32, x (N,3) f32 colours, heights (N,1) f32, y (N,) i64)."""
    rng = np.random.default_rng(seed + int(sample_id))
    rects = _room_surfaces(rng)
    area = np.array([np.linalg.norm(np.cross(eu, ev)) for _, eu, ev, _ in rects])
    # oversample surfaces ~6 points per voxel, then keep one point per voxel
    need = int(n_points * (3.0 if not duplicates else 0.8))
    dens = max(6.0 / voxel_size ** 2, 1.0)
    while True:
        pts, lab = [], []
        for (o, eu, ev, cid), a in zip(rects, area):
            k = max(int(a * dens), 1)
            uv = rng.uniform(size=(k, 2))
            pts.append(o + uv[:, :1] * eu + uv[:, 1:] * ev)
            # one class per ~1.5 m patch of a surface: spatially coherent labels with a realistic
            # share (10-25 %) of points whose neighbourhood straddles a class boundary
            pu = np.floor(uv[:, 0] * np.linalg.norm(eu) / 1.5).astype(np.int64)
            pv = np.floor(uv[:, 1] * np.linalg.norm(ev) / 1.5).astype(np.int64)
            lab.append((cid * 7 + pu * 3 + pv * 5) % num_classes)
        pts = np.concatenate(pts).astype(np.float64)
        lab = np.concatenate(lab)
        pts += rng.normal(scale=voxel_size * 0.05, size=pts.shape)
        key = np.floor(pts / voxel_size).astype(np.int64)
        key -= key.min(0)
        flat = (key[:, 0] * (key[:, 1].max() + 1) + key[:, 1]) * (key[:, 2].max() + 1) + key[:, 2]
        order = rng.permutation(len(flat))
        _, first = np.unique(flat[order], return_index=True)
        keep = order[first]
        pts, lab = pts[keep], lab[keep]
        if len(pts) >= need or duplicates or voxel_size < 1e-3:
            break
        voxel_size *= 0.8  # room too small for the request: refine the voxels
        dens = 6.0 / voxel_size ** 2
    # crop: the n_points nearest to a random centre (data_util.py:157-160)
    centre = pts[rng.integers(len(pts))]
    d = ((pts - centre) ** 2).sum(1)
    if duplicates:  # a room with fewer voxels than voxel_max: keep 80 % distinct points, pad the rest
        near = np.argsort(d, kind="stable")[:max(int(n_points * 0.8), 1)]
        pts, lab, d = pts[near], lab[near], d[near]
    if len(pts) >= n_points:
        sel = np.argsort(d, kind="stable")[:n_points]
    else:  # pad by repetition (data_util.py:161-167) -> exact duplicate points
        sel = np.concatenate([np.arange(len(pts)), rng.integers(len(pts), size=n_points - len(pts))])
    sel = sel[rng.permutation(len(sel))]
    pts, lab = pts[sel], lab[sel]
    pts = pts - pts.min(0)  # data_util.py:173
    pos = pts.astype(np.float32)
    col = rng.uniform(size=(n_points, 3)).astype(np.float32)
    if ignore_frac > 0:
        m = rng.uniform(size=n_points) < ignore_frac
        lab = lab.copy(); lab[m] = ignore_index
    return {"pos": pos, "x": col, "heights": pos[:, 2:3].copy(), "y": lab}


def make_batch(batch, n_points, first_id=0, **kw):
    """Batch dict in the layout main_AA.py hands to the model after
    get_features_by_keys(data, 'x,heights') (dataset/data_util.py:177-189):
    pos (B,N,3) f32, x (B,4,N) f32, y (B,N) i64 -- numpy arrays."""
    scenes = [make_scene(first_id + i, n_points, **kw) for i in range(batch)]
    pos = np.stack([s["pos"] for s in scenes])
    feat = np.stack([np.concatenate([s["x"], s["heights"]], 1).T for s in scenes])
    y = np.stack([s["y"] for s in scenes])
    return {"pos": np.ascontiguousarray(pos), "x": np.ascontiguousarray(feat.astype(np.float32)),
            "y": np.ascontiguousarray(y)}
