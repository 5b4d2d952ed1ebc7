"""ORACLE -- generates tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python oracle/gen_golden.py            # rewrites every fixture

The reference's own Python layer -- openpoints.models (BaseSeg_AMContrast3D, PointNeXt
encoder/decoder, SegHead), openpoints.models.layers (QueryAndGroup, three_interpolation,
furthest_point_sample, ...), openpoints.loss.CrossEntropyAce and
openpoints.AMContrast3D (ContrastHead, ambiguity_function, get_subscene_label_CBL) -- is
imported read-only from /root/reference (oracle/refshim.py) and executed on the CPU with
the C restatement of its CUDA kernels (oracle/pointops_ref.c) underneath.  What is stored
are inputs and the reference's outputs: data, never code.  The fixtures pin
  * the oracle's model/loss restatement (oracle/model_ref.py)   [tests, CPU]
  * the product (amcontrast3d_amd + HIP kernels)                [tests, GPU]
The native kernels themselves cannot be run from the reference (CUDA only); their
restatement is pinned by the brute-force cross-checks in tests/test_oracle_ops.py.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import refshim  # noqa: E402

refshim.load_reference()

from openpoints.loss import build_criterion_from_cfg  # noqa: E402  (reference)
from openpoints.models import build_model_from_cfg  # noqa: E402  (reference)
from openpoints.models.layers import (furthest_point_sample, three_interpolation, three_nn)  # noqa: E402
from openpoints.models.layers.group import QueryAndGroup, ball_query, grouping_operation  # noqa: E402
from openpoints.cpp.pointops.functions import pointops  # noqa: E402
from openpoints.utils import EasyConfig  # noqa: E402
from openpoints.AMContrast3D.AEF.ambiguity import ambiguity_function  # noqa: E402

from amcontrast3d_amd import configs, synthetic  # noqa: E402  (plain dict/numpy helpers only)

torch.set_num_threads(8)


def cfg_of(d):
    c = EasyConfig()
    c.update(d)
    return c


def tensor_batch(batch):
    return {"pos": torch.from_numpy(batch["pos"]), "x": torch.from_numpy(batch["x"]), "y": torch.from_numpy(batch["y"])}


def param_checksums(model):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in model.state_dict().items()
            if v.dtype.is_floating_point}


def run_model_case(name, variant, B, N, num_classes=13, in_channels=4, dataset="s3dis", ignore_index=None,
                   ignore_frac=0.0, store_weights=False, grad_keys=(), voxel_size=0.04, **model_kw):
    """One forward + loss + backward of the reference; everything a parity test needs goes in the .npz."""
    torch.manual_seed(0)
    mcfg = cfg_of(configs.model_cfg(variant, num_classes=num_classes, in_channels=in_channels, dropout=0, **model_kw))
    model = build_model_from_cfg(mcfg)
    model.train()
    criterion = build_criterion_from_cfg(cfg_of(configs.criterion_cfg()))
    aargs = cfg_of(configs.ambiguity_args(dataset))

    nb = synthetic.make_batch(B, N, first_id=100, num_classes=num_classes, ignore_frac=ignore_frac,
                              voxel_size=voxel_size)
    if in_channels == 7:  # ScanNet feature_keys 'pos,x,heights' (cfgs/scannet/default.yaml:21)
        nb["x"] = np.ascontiguousarray(np.concatenate([nb["pos"].transpose(0, 2, 1), nb["x"]], 1))
    data = tensor_batch(nb)
    target = data["y"]

    out = {"pos": nb["pos"], "x": nb["x"], "y": nb["y"]}
    sums = param_checksums(model)
    if store_weights:
        for k, v in model.state_dict().items():
            out["w/" + k] = v.numpy().copy()

    logits, stage = model(data)
    loss = criterion(logits, target, stage, num_classes, ignore_index, aargs)
    loss.backward()

    out["logits"] = logits.detach().numpy()
    out["loss"] = np.float64(loss.item())
    ce = torch.nn.CrossEntropyLoss()(logits.transpose(1, 2).reshape(-1, logits.shape[1]), target.flatten())
    out["loss_ce"] = np.float64(ce.item())
    head = criterion.contrast_head
    for i in range(aargs.stages_num):
        li, _, ai = head.main_contrast(aargs.stages, i, stage, target.flatten(), num_classes, ignore_index, aargs)
        out[f"contrast/{i}"] = np.float64(li.item())
        out[f"ambiguity/{i}"] = ai.detach().numpy()
        out[f"p_out/{i}"] = stage["up"][i]["p_out"].detach().numpy()
        out[f"f_out/{i}"] = stage["up"][i]["f_out"].detach().numpy()
    grads = {k: p.grad for k, p in model.named_parameters()}
    out["grad_abs_sum"] = np.float64(sum(float(g.double().abs().sum()) for g in grads.values()))
    for k in grad_keys:
        out["g/" + k] = grads[k].numpy().copy()
    gn = {k: float(g.double().norm()) for k, g in grads.items()}
    meta = {"variant": variant, "B": B, "N": N, "num_classes": num_classes, "in_channels": in_channels,
            "dataset": dataset, "ignore_index": ignore_index, "ignore_frac": ignore_frac, "voxel_size": voxel_size,
            "model_kw": model_kw,
            "param_checksums": sums, "grad_norms": gn, "torch": torch.__version__}
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} ce={ce.item():.6f} "
          + " ".join(f"c{i}={out[f'contrast/{i}']:.5f}" for i in range(aargs.stages_num)),
          f"({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e6:.2f} MB)")


def run_mm_case(name, variant, B, N, num_classes=13, in_channels=4, dataset="s3dis", store_weights=True, grad_keys=(),
                **model_kw):
    """AMContrast3D++ (BaseSeg_M_AMContrast3D + CrossEntropyAcePre): one forward + loss + backward of the reference,
    loss = segmentation + regression as examples/segmentation/main_MM.py:404-410 combines them."""
    torch.manual_seed(0)
    mcfg = cfg_of(configs.model_cfg_mm(variant, num_classes=num_classes, in_channels=in_channels, dropout=0, dataset=dataset,
                                       **model_kw))
    model = build_model_from_cfg(mcfg)
    model.train()
    criterion = build_criterion_from_cfg(cfg_of(configs.criterion_cfg_mm()))
    aargs = cfg_of(configs.ambiguity_args_mm(dataset))
    nb = synthetic.make_batch(B, N, first_id=300, num_classes=num_classes)
    data = tensor_batch(nb)
    target = data["y"]
    out = {"pos": nb["pos"], "x": nb["x"], "y": nb["y"]}
    sums = param_checksums(model)
    if store_weights:
        for k, v in model.state_dict().items():
            out["w/" + k] = v.numpy().copy()
    logits, stage, refine = model(data)
    seg, ce, contrast, reg = criterion(logits, target, stage, num_classes, None, aargs)
    loss = seg + reg
    loss.backward()
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.float64(loss.item())
    out["loss_ce"], out["loss_contrast"], out["loss_reg"] = (np.float64(v.item()) for v in (ce, contrast, reg))
    out["refine_rate"] = np.float64(refine)
    for i in range(4):
        out[f"apm/{i}"] = stage["ambiguity"][i].detach().numpy()
        out[f"f_out/{i}"] = stage["up"][i]["f_out"].detach().numpy()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    out["grad_abs_sum"] = np.float64(sum(float(g.double().abs().sum()) for g in grads.values()))
    for k in grad_keys:
        out["g/" + k] = grads[k].numpy().copy()
    meta = {"variant": variant, "B": B, "N": N, "num_classes": num_classes, "in_channels": in_channels, "dataset": dataset,
            "model_kw": model_kw, "param_checksums": sums, "grad_norms": {k: float(g.double().norm()) for k, g in grads.items()},
            "state_keys": list(model.state_dict().keys()), "torch": torch.__version__}
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} ce={ce.item():.6f} contrast={contrast.item():.6f} reg={reg.item():.6f} "
          f"refine={refine:.3f}% ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e6:.2f} MB)")


def run_ops_case():
    """The reference's Python wrappers on small clouds, including tie-heavy ones."""
    rng = np.random.default_rng(7)
    out = {}
    # -- clouds: a synthetic room crop, an integer lattice (many exact distance ties), duplicates --
    room = synthetic.make_batch(2, 512, first_id=7)["pos"]
    lattice = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    lattice = np.stack([lattice[rng.permutation(512)] for _ in range(2)]).astype(np.float32) * 0.25
    dup = room.copy()
    dup[:, 256:] = dup[:, :256]  # every point appears twice
    for tag, cloud, radius in (("room", room, 0.2), ("lattice", lattice, 0.5), ("dup", dup, 0.2)):
        xyz = torch.from_numpy(np.ascontiguousarray(cloud))
        B, N, _ = xyz.shape
        out[f"{tag}/xyz"] = cloud
        fidx = furthest_point_sample(xyz, N // 4)
        out[f"{tag}/fps"] = fidx.numpy()
        new_xyz = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        out[f"{tag}/ball"] = ball_query(radius, 32, xyz, new_xyz).numpy()
        out[f"{tag}/ball_radius"] = np.float32(radius)
        feats = torch.from_numpy(rng.standard_normal((B, 5, N)).astype(np.float32))
        out[f"{tag}/feats"] = feats.numpy()
        dp, fj = QueryAndGroup(radius, 32, normalize_dp=True)(new_xyz, xyz, feats)
        out[f"{tag}/group_dp"] = dp.numpy()
        out[f"{tag}/group_fj"] = fj.numpy()
        d, i3 = three_nn(xyz, new_xyz)
        out[f"{tag}/three_nn_dist"] = d.numpy()
        out[f"{tag}/three_nn_idx"] = i3.numpy()
        cf = torch.from_numpy(rng.standard_normal((B, 5, N // 4)).astype(np.float32))
        out[f"{tag}/coarse_feats"] = cf.numpy()
        out[f"{tag}/interp"] = three_interpolation(xyz, new_xyz, cf).numpy()
        # k-NN over the flattened batch as ONE segment (pointnext_AA.py:459-462) and as B segments
        flat = xyz.reshape(-1, 3).contiguous()
        one = torch.tensor([B * N], dtype=torch.int32)
        per = torch.tensor([N * (b + 1) for b in range(B)], dtype=torch.int32)
        for seg_tag, off in (("one", one), ("per", per)):
            idx, dist = pointops.knnquery(24, flat, flat, off, off)
            out[f"{tag}/knn24_{seg_tag}_idx"] = idx.numpy()
            out[f"{tag}/knn24_{seg_tag}_dist"] = dist.numpy()
        q = new_xyz.reshape(-1, 3).contiguous()
        qone = torch.tensor([q.shape[0]], dtype=torch.int32)
        idx, dist = pointops.knnquery(64, flat, q, one, qone)
        out[f"{tag}/knn64_idx"] = idx.numpy()
        out[f"{tag}/knn64_dist"] = dist.numpy()
    # -- grouping gradient (atomics in the reference; ascending order in the restatement) --
    xyz = torch.from_numpy(np.ascontiguousarray(room))
    fidx = furthest_point_sample(xyz, 128)
    new_xyz = torch.gather(xyz, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = ball_query(0.2, 32, xyz, new_xyz)
    feats = torch.from_numpy(rng.standard_normal((2, 6, 512)).astype(np.float32)).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((2, 6, 128, 32)).astype(np.float32))
    grouping_operation(feats, idx).backward(g)
    out["grad/feats"], out["grad/g"], out["grad/idx"] = feats.detach().numpy(), g.numpy(), idx.numpy()
    out["grad/group_grad"] = feats.grad.numpy()
    # -- ambiguity function on a labelled room (reference loop :32-35 included) --
    sc = synthetic.make_scene(3, 3000)
    p = torch.from_numpy(sc["pos"])
    lab = torch.from_numpy(sc["y"])
    off = torch.tensor([3000], dtype=torch.int32)
    nidx, _ = pointops.knnquery(24, p, p, off, off)
    nidx = nidx[:, 1:].contiguous()
    posmask = lab[:, None] == lab[nidx.long()]
    a, shares = ambiguity_function(p, posmask, 23, nidx, "Method2", 0.04, False, 0.5)
    out["amb/p"], out["amb/label"], out["amb/nidx"] = sc["pos"], sc["y"], nidx.numpy()
    out["amb/a"] = a.numpy()
    out["amb/shares"] = np.array(shares, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **out)
    print("ops_small:", len(out), "arrays,", os.path.getsize(os.path.join(OUT, "ops_small.npz")) / 1e6, "MB")


def run_state_keys():
    keys = {}
    for variant in ("S", "B", "L", "XL"):
        model = build_model_from_cfg(cfg_of(configs.model_cfg(variant)))
        keys[variant] = {k: list(v.shape) for k, v in model.state_dict().items()}
        keys[variant + "_nparams"] = sum(p.numel() for p in model.parameters())
    sc = build_model_from_cfg(cfg_of(configs.model_cfg("S", num_classes=20, in_channels=7, global_feat="max")))
    keys["S_scannet"] = {k: list(v.shape) for k, v in sc.state_dict().items()}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print("state_keys:", {k: (v if isinstance(v, int) else len(v)) for k, v in keys.items()})


def run_eval_case(name="eval_w8_room", n_room=5000, voxel=0.06, width=8, nsample=24, num_classes=13, ignore_index=None):
    """Whole-room test of the reference in eval mode (examples/segmentation/main_AA.py:556-684 for one cloud):
    the reference's model, voxelize, posmask_searching, ConfusionMatrix and get_mious are executed; the loop glue of
    main_AA.py (not importable here: wandb / torch_scatter) and torch_scatter's mean are restated in this function."""
    from openpoints.AMContrast3D.metrics import posmask_searching
    from openpoints.dataset.data_util import voxelize
    from openpoints.utils import ConfusionMatrix, get_mious
    torch.manual_seed(0)
    mcfg = cfg_of(configs.model_cfg("S", num_classes=num_classes, in_channels=4, dropout=0.5, width=width))
    model = build_model_from_cfg(mcfg)
    model.train()
    with torch.no_grad():  # two training-mode passes so that the running statistics are not the initial 0 / 1
        for k in range(2):
            model(tensor_batch(synthetic.make_batch(2, 1024, first_id=500 + 2 * k, num_classes=num_classes)))
    model.eval()
    room = synthetic.make_batch(1, n_room, first_id=300, num_classes=num_classes)
    coord = room["pos"][0].astype(np.float32)
    coord = coord - coord.min(0)
    feat = room["x"][0, :3].T.copy()
    label_np = room["y"][0].astype(np.int64)
    # 4 % label noise: isolated labels are what makes a point "inner" under the loop's 0 < n+ < nsample test
    noise = np.random.default_rng(7)
    flip = noise.random(n_room) < 0.04
    label_np[flip] = noise.integers(0, num_classes, int(flip.sum()))
    label = torch.from_numpy(label_np)

    np.random.seed(0)
    idx_sort, voxel_idx, count = voxelize(coord, voxel, mode=1)
    parts = []
    for i in range(count.max()):  # main_AA.py:109-113
        part = idx_sort[np.cumsum(np.insert(count, 0, 0)[0:-1]) + i % count]
        np.random.shuffle(part)
        parts.append(part)

    out = {"coord": coord, "feat": feat, "label": label_np, "voxel": np.float64(voxel), "nsample": np.int64(nsample)}
    for k, v in model.state_dict().items():
        out["w/" + k] = v.numpy().copy()
    cm, cm_b, cm_i = (ConfusionMatrix(num_classes=num_classes, ignore_index=ignore_index) for _ in range(3))
    all_logits, lb, li, tb, ti = [], [], [], [], []
    with torch.no_grad():
        for j, part in enumerate(parts):
            cp = coord[part]
            cp = cp - cp.min(0)
            pos = torch.from_numpy(cp).unsqueeze(0)
            x = torch.cat([torch.from_numpy(feat[part]), pos[0, :, 2:3]], 1).t().contiguous().unsqueeze(0)
            logits, _ = model({"pos": pos, "x": x})
            all_logits.append(logits)
            label_part = label[part]
            posmask, _ = posmask_searching(pos.squeeze(), label_part, nsample, num_classes, ignore_index)
            point_mask = torch.sum(posmask.int(), -1)
            boundary = torch.logical_and(0 < point_mask, point_mask < nsample).unsqueeze(0)
            pred_part = logits.argmax(dim=1)
            lb.append(torch.masked_select(pred_part, boundary)); li.append(torch.masked_select(pred_part, ~boundary))
            tb.append(torch.masked_select(label_part.unsqueeze(0), boundary))
            ti.append(torch.masked_select(label_part.unsqueeze(0), ~boundary))
            out[f"part/{j}"] = part.astype(np.int32)
            if j in (0, len(parts) - 1):  # per-part logits of the first and last sub-cloud only (fixture size)
                out[f"logits/{j}"] = logits[0].numpy().copy()
            out[f"boundary/{j}"] = np.packbits(boundary[0].numpy())
    flat = torch.cat(all_logits, 0).transpose(1, 2).reshape(-1, num_classes) if len({lg.shape[2] for lg in all_logits}) == 1 \
        else torch.cat([lg.transpose(1, 2).reshape(-1, num_classes) for lg in all_logits], 0)
    index = torch.from_numpy(np.hstack(parts))
    voted = torch.zeros(len(label), num_classes).index_add_(0, index, flat)  # torch_scatter 'mean': sum / max(count, 1)
    voted = voted / torch.bincount(index, minlength=len(label)).clamp(min=1).unsqueeze(1)
    pred = voted.argmax(dim=1)
    cm.update(pred, label.clone())
    cm_b.update(torch.cat(lb), torch.cat(tb))
    cm_i.update(torch.cat(li), torch.cat(ti))
    out["voted"], out["pred"] = voted.numpy(), pred.numpy()
    for tag, m in (("all", cm), ("boundary", cm_b), ("inner", cm_i)):
        out[f"cm/{tag}"] = m.value.numpy().copy()
        miou, macc, oa, ious, accs = get_mious(m.tp, m.union, m.count)
        out[f"mious/{tag}"] = np.array([miou, macc, oa], dtype=np.float64)
        out[f"ious/{tag}"], out[f"accs/{tag}"] = ious, accs
        out[f"all_metrics/{tag}"] = np.array(m.all_metrics()[:3], dtype=np.float64)
    # ---- accuracy per ambiguity level (main_AA.py:686-702 with ambiguity_args.action) on the whole cloud ----------
    from openpoints.AMContrast3D.metrics import ambiguity_metrics
    p_all = torch.from_numpy(coord)
    posmask_test, nidx_test = posmask_searching(p_all, label, nsample, num_classes, ignore_index)
    cms = [ConfusionMatrix(num_classes=num_classes, ignore_index=ignore_index) for _ in range(5)]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints every table
        a_soft, ratio, a_count, ratio_lsh, cls, l_miou, l_macc, l_oa, l_count = ambiguity_metrics(
            p_all, label, pred, posmask_test, nsample, nidx_test, "Method2", 0.04, False, *cms, 0.5)
    out["amb/a"] = a_soft.numpy().astype(np.float32)
    for j, m in enumerate(cms):
        out[f"amb/cm/{j}"] = m.value.numpy().copy()
    amb = {"ratio": {str(k): v for k, v in ratio.items()}, "count": list(a_count), "ratio_low_semi_high": ratio_lsh,
           "cls": {str(k): v for k, v in cls.items()}, "miou": l_miou, "macc": l_macc, "oa": l_oa, "count_per_class": l_count}
    meta = {"num_classes": num_classes, "ignore_index": ignore_index, "width": width, "n_room": n_room,
            "parts": len(parts), "torch": torch.__version__, "ambiguity_metrics": amb}
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: parts={[len(p) for p in parts]} mIoU={out['mious/all'][0]:.3f} boundary pts={int(cm_b.value.sum())} "
          f"inner pts={int(cm_i.value.sum())} ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e6:.2f} MB)")


def run_input_case(name="input_room", n_base=12000, copies=3, voxel=0.04, voxel_max=1000):
    """Input pipeline of the S3DIS loader (dataset/s3dis/s3dis.py:122-144 -> dataset/data_util.py:92-174): the reference's
    own fnv_hash_vec / voxelize / crop_pc run on a raw (several points per voxel) synthetic room.  numpy's argsort is
    not stable, so the order of points INSIDE a voxel (idx_sort) is whatever numpy's sort made of it; recorded as is,
    compared as sets per voxel."""
    from openpoints.dataset.data_util import crop_pc, fnv_hash_vec, voxelize
    room = synthetic.make_batch(1, n_base, first_id=700, voxel_size=0.02)
    rng = np.random.default_rng(11)
    base = room["pos"][0].astype(np.float32)
    coord = np.concatenate([base + rng.uniform(-0.015, 0.015, base.shape).astype(np.float32) for _ in range(copies)], 0)
    feat = np.concatenate([room["x"][0, :3].T] * copies, 0).astype(np.float32)
    label = np.concatenate([room["y"][0]] * copies, 0).astype(np.int64)
    perm = rng.permutation(len(coord))
    coord, feat, label = coord[perm], feat[perm], label[perm]
    coord = coord - coord.min(0)
    key = fnv_hash_vec(np.floor(coord / np.array(voxel)))
    idx_sort, voxel_idx, count = voxelize(coord, voxel, mode=1)
    np.random.seed(3)
    rnd = np.random.randint(0, count.max(), count.size)   # the draw voxelize(mode=0) makes (data_util.py:138-139)
    np.random.seed(3)
    idx_unique = voxelize(coord, voxel, mode=0)
    assert np.array_equal(idx_unique, idx_sort[np.cumsum(np.insert(count, 0, 0)[0:-1]) + rnd % count])
    # the crop stage alone on the voxelised cloud (validation split: centre point N // 2, no shuffle)
    cv, fv, lv = coord[idx_unique].copy(), feat[idx_unique].copy(), label[idx_unique].copy()
    d2 = np.sum(np.square(cv - cv[len(lv) // 2]), 1)
    crop_idx = np.argsort(d2)[:voxel_max]
    c_out, f_out, l_out = crop_pc(cv.copy(), fv.copy(), lv.copy(), "val", voxel, voxel_max, downsample=False, variable=True,
                                  shuffle=False)
    assert np.array_equal(c_out, (cv[crop_idx] - cv[crop_idx].min(0)).astype(np.float32))
    out = {"coord": coord, "feat": feat, "label": label, "voxel": np.float64(voxel), "voxel_max": np.int64(voxel_max),
           "key": key, "idx_sort": idx_sort.astype(np.int64), "voxel_idx": voxel_idx.astype(np.int64),
           "count": count.astype(np.int64), "rnd": rnd.astype(np.int64), "idx_unique": idx_unique.astype(np.int64),
           "d2": d2.astype(np.float32), "crop_idx": crop_idx.astype(np.int64), "crop_coord": c_out, "crop_feat": f_out,
           "crop_label": l_out.astype(np.int64),
           "meta": json.dumps({"numpy": np.__version__, "note": "coord / np.array(voxel) is a float64 division under numpy 2 "
                               "(NEP 50: a 0-d array is not a weak scalar)"})}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "raw points", len(coord), "voxels", len(count), "max per voxel", int(count.max()), "crop", len(crop_idx))


def run_augment_case(name="augment_s3dis", n=3000):
    """The training transform chain of cfgs/s3dis/default.yaml:33-43, run with the reference's OWN classes
    (transforms/point_transform_cpu.py:8-19,192-209, transforms/point_transformer_gpu.py:70-89,135-164,216-229,267-311,373-409)
    on one cropped cloud, exactly as S3DIS.__getitem__ calls it (dataset/s3dis/s3dis.py:136-143, including `heights`, which
    reads the numpy array handed to the transforms -- untouched by them: PointsToTensor copies before torch.from_numpy).  Every random number the classes draw is logged by wrapping the generator functions they call; stored are
    the inputs, the draws and the outputs.  Two cases: auto-contrast and colour drop taken / not taken."""
    import collections
    import collections.abc
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable  # removed from `collections` in Python 3.10; the reference's image had 3.7
    from openpoints.transforms import point_transform_cpu as T_cpu, point_transformer_gpu as T_gpu
    room = synthetic.make_batch(1, n, first_id=811)
    coord0 = (room["pos"][0] - room["pos"][0].min(0)).astype(np.float32)
    feat0 = (room["x"][0, :3].T * 255.0).astype(np.float32)
    out = {"coord": coord0, "feat": feat0}
    kw = dict(color_drop=0.2, gravity_dim=2, scale=[0.9, 1.1], angle=[0, 0, 1], jitter_sigma=0.005, jitter_clip=0.02)
    for tag, p_contrast, p_drop, seed in (("a", 1.0, 1.0, 5), ("b", 0.0, 0.0, 6)):
        log = []
        orig = (torch.rand, torch.randn_like, np.random.rand, np.random.uniform)
        torch.rand = lambda *a, **k: (lambda t: (log.append(("torch.rand", t.clone().numpy())), t)[1])(orig[0](*a, **k))
        torch.randn_like = lambda *a, **k: (lambda t: (log.append(("torch.randn_like", t.clone().numpy())), t)[1])(orig[1](*a, **k))
        np.random.rand = lambda *a: (lambda v: (log.append(("np.rand", np.array(v))), v)[1])(orig[2](*a))
        np.random.uniform = lambda *a, **k: (lambda v: (log.append(("np.uniform", np.array(v))), v)[1])(orig[3](*a, **k))
        try:
            np.random.seed(seed)
            torch.manual_seed(seed)
            chain = [T_cpu.ChromaticAutoContrast(p=p_contrast), T_cpu.PointsToTensor(), T_gpu.PointCloudScaling(**kw),
                     T_gpu.PointCloudXYZAlign(**kw), T_gpu.PointCloudRotation(**kw), T_gpu.PointCloudJitter(**kw),
                     T_gpu.ChromaticDropGPU(color_drop=p_drop), T_gpu.ChromaticNormalize()]
            coord, feat = coord0.copy(), feat0.copy()
            data = {"pos": coord, "x": feat}
            for t in chain:
                data = t(data)
            heights = torch.from_numpy(coord[:, 2:3].astype(np.float32))  # s3dis.py:141-142
        finally:
            torch.rand, torch.randn_like, np.random.rand, np.random.uniform = orig
        kinds = [k for k, _ in log]
        # the order the classes draw in: [contrast u, (blend)], scale (3), theta x / y / z, noise (n,3), drop u
        i = 0
        out[f"{tag}/contrast_u"] = log[i][1]; i += 1
        if p_contrast >= 1.0:
            out[f"{tag}/blend"] = log[i][1]; i += 1
        assert kinds[i] == "torch.rand" and log[i][1].shape == (3,)
        out[f"{tag}/scale_u"] = log[i][1]; i += 1
        assert kinds[i:i + 3] == ["np.uniform"] * 3
        out[f"{tag}/theta"] = np.array([log[i][1], log[i + 1][1], log[i + 2][1]], dtype=np.float64); i += 3
        assert kinds[i] == "torch.randn_like"
        out[f"{tag}/noise"] = log[i][1]; i += 1
        out[f"{tag}/drop_u"] = log[i][1]; i += 1
        assert i == len(log), kinds
        out[f"{tag}/p_contrast"], out[f"{tag}/p_drop"] = np.float64(p_contrast), np.float64(p_drop)
        out[f"{tag}/pos"], out[f"{tag}/x"], out[f"{tag}/heights"] = data["pos"].numpy(), data["x"].numpy(), heights.numpy()
        print(name, tag, "draws", kinds, "pos range", float(data["pos"].min()), float(data["pos"].max()))
    out["meta"] = json.dumps(kw)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def run_mm_cases():
    # AMContrast3D++ with a narrow S-shaped backbone and stored weights: APM towers, masked refinement (DualMasks,
    # threshold lowered so that a good share of the points is refined at seed-0 init), three-term loss
    run_mm_case("model_mm_w8_b2_n2048", "S", 2, 2048, width=8, threshold=0.5,
                grad_keys=["APM.layer_0.0.weight", "APM.layer_3.20.weight", "encoder.encoder.1.0.convs.0.0.weight",
                           "decoder.decoder.0.0.convs.0.0.weight", "head.head.1.0.weight"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "mm":
        run_mm_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        run_eval_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "input":
        run_input_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        run_augment_case()
        sys.exit(0)
    run_state_keys()
    run_ops_case()
    G = ["encoder.encoder.0.0.convs.0.0.weight", "encoder.encoder.1.0.convs.0.0.weight",
         "encoder.encoder.4.0.convs.1.1.weight", "decoder.decoder.0.0.convs.0.0.weight", "head.head.1.0.weight"]
    # PointNeXt-S, seed-0 init (weights re-created from the seed; checksums stored)
    run_model_case("model_S_b2_n2048", "S", 2, 2048, grad_keys=G)
    # BASELINE config 1 shape: S, B=2, N=4096
    run_model_case("model_S_b2_n4096", "S", 2, 4096, grad_keys=G[:2])
    # narrow model with InvResMLP blocks and stored weights (independent of RNG streams)
    run_model_case("model_w8_blocks_b2_n1024", "L", 2, 1024, width=8, blocks=[1, 2, 2, 1, 1], store_weights=True,
                   grad_keys=["encoder.encoder.1.1.convs.convs.0.0.weight", "encoder.encoder.2.1.pwconv.1.0.weight"])
    # ScanNet-shaped: 20 classes + ignore_index -100, 7 input channels, global max feature in the head
    run_model_case("model_S_scannet_b2_n2048", "S", 2, 2048, num_classes=20, in_channels=7, dataset="scannet",
                   ignore_index=-100, ignore_frac=0.05, voxel_size=0.02, global_feat="max", grad_keys=G[:1])
    run_mm_cases()
    run_eval_case()
    run_input_case()
    run_augment_case()
