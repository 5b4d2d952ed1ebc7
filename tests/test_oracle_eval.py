"""Evaluation path, CPU part: the oracle's restatement (oracle/eval_ref.py) and the
product's host-side pieces (openpoints.utils.metrics, amcontrast3d_amd.evaluate.voxel_parts / scatter_mean) against
tests/golden/eval_w8_room.npz, which was recorded from the reference's own eval-mode model, voxelize,
posmask_searching, ConfusionMatrix and get_mious (oracle/gen_golden.py eval).  Integer results bit-exact; logits 1e-4."""
import numpy as np
import pytest
import torch

from amcontrast3d_amd import configs
from conftest import load_golden
from oracle import eval_ref

NAME = "eval_w8_room"


def setup():
    g = load_golden(NAME)
    m = g["meta"]
    parts = [g[f"part/{j}"].astype(np.int64) for j in range(m["parts"])]
    boundary = [np.unpackbits(g[f"boundary/{j}"])[:len(parts[j])].astype(bool) for j in range(m["parts"])]
    return g, m, parts, boundary


def near_tie_free(voted, tol):
    top2 = np.sort(voted, axis=1)[:, -2:]
    return (top2[:, 1] - top2[:, 0]) > tol


def test_confusion_and_mious_restatement():
    g, m, parts, _ = setup()
    cm = eval_ref.confusion(g["pred"], g["label"], m["num_classes"], m["ignore_index"])
    assert np.array_equal(cm, g["cm/all"])
    for tag in ("all", "boundary", "inner"):
        mat = g[f"cm/{tag}"]
        miou, macc, oa, ious, accs = eval_ref.get_mious(*eval_ref.tp_union_count(mat))
        np.testing.assert_allclose([miou, macc, oa], g[f"mious/{tag}"], rtol=2e-6)
        np.testing.assert_allclose(ious, g[f"ious/{tag}"], rtol=2e-6)
        np.testing.assert_allclose(accs, g[f"accs/{tag}"], rtol=2e-6)
        np.testing.assert_allclose(eval_ref.all_metrics(mat)[:3], g[f"all_metrics/{tag}"], rtol=2e-6)


def test_confusion_ignore_index():
    pred = np.array([0, 1, 2, 2, 1, 0])
    true = np.array([0, 1, -100, 2, -100, 1])
    cm = eval_ref.confusion(pred, true, 3, ignore_index=-100)
    assert cm.sum() == 4 and cm[0, 0] == 1 and cm[1, 1] == 1 and cm[2, 2] == 1 and cm[1, 0] == 1


def test_voxel_partition_restatement():
    g, m, parts, _ = setup()
    mine = eval_ref.voxel_parts(g["coord"], float(g["voxel"]))
    assert len(mine) == len(parts)
    for a, b in zip(mine, parts):  # the reference shuffles each part with the global numpy RNG: compare as sets
        assert np.array_equal(np.sort(a), np.sort(b))
    assert np.array_equal(np.unique(np.hstack(mine)), np.arange(len(g["coord"])))


def test_boundary_restatement():
    g, m, parts, boundary = setup()
    for j in (0, len(parts) - 1):
        cp = g["coord"][parts[j]]
        cp = cp - cp.min(0)
        b = eval_ref.boundary_mask(cp, g["label"][parts[j]], int(g["nsample"]), m["num_classes"], m["ignore_index"])
        assert np.array_equal(b, boundary[j])
        assert 0 < b.sum() < len(b)  # both kinds present (the fixture has 4 % label noise)


def test_whole_cloud_restatement():
    g, m, parts, boundary = setup()
    cfg = configs.model_cfg("S", num_classes=m["num_classes"], in_channels=4, dropout=0, width=m["width"])
    sd = {k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("w/")}
    torch.set_num_threads(8)
    r = eval_ref.test_cloud(sd, cfg, g["coord"], g["feat"], g["label"], parts, m["num_classes"], m["ignore_index"],
                            int(g["nsample"]))
    scale = max(1.0, float(np.abs(g["voted"]).max()))
    for j in (0, len(parts) - 1):
        assert np.abs(r["logits_parts"][j] - g[f"logits/{j}"]).max() <= 1e-4 * scale
    assert np.abs(r["voted"] - g["voted"]).max() <= 1e-4 * scale
    clear = near_tie_free(g["voted"], 2e-4 * scale)
    assert clear.mean() > 0.9 and np.array_equal(r["pred"][clear], g["pred"][clear])
    slack = int((~clear).sum())
    assert np.abs(r["cm"] - g["cm/all"]).sum() <= 2 * slack
    # boundary / inner matrices use per-part predictions: same slack argument on the totals, exact on the row sums
    for tag in ("cm_b", "cm_i"):
        ref = g["cm/boundary" if tag == "cm_b" else "cm/inner"]
        assert np.array_equal(r[tag].sum(1), ref.sum(1))


def test_product_metrics_classes_on_cpu():
    """openpoints.utils.ConfusionMatrix / get_mious of the product are plain torch integer code: run them here."""
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.utils import AverageMeter, ConfusionMatrix, get_mious
    g, m, parts, _ = setup()
    cm = ConfusionMatrix(num_classes=m["num_classes"], ignore_index=m["ignore_index"])
    pred, label = torch.from_numpy(g["pred"]), torch.from_numpy(g["label"])
    half = len(label) // 2
    cm.update(pred[:half], label[:half])
    cm.update(pred[half:], label[half:])
    assert np.array_equal(cm.value.numpy(), g["cm/all"])
    miou, macc, oa, ious, accs = get_mious(cm.tp, cm.union, cm.count)
    np.testing.assert_allclose([miou, macc, oa], g["mious/all"], rtol=1e-6)
    np.testing.assert_allclose(ious, g["ious/all"], rtol=1e-6)
    np.testing.assert_allclose(cm.all_metrics()[:3], g["all_metrics/all"], rtol=1e-6)
    assert int(cm.total) == len(label) and int((cm.tp + cm.fp + cm.fn - cm.union).abs().sum()) == 0
    assert int((cm.tn + cm.tp + cm.fp + cm.fn - cm.total).abs().sum()) == 0
    # ignore_index: dropped, and the caller's tensors stay untouched
    cm2 = ConfusionMatrix(num_classes=3, ignore_index=-100)
    p, t = torch.tensor([0, 1, 2, 2, 1, 0]), torch.tensor([0, 1, -100, 2, -100, 1])
    cm2.update(p, t)
    assert int(cm2.total) == 4 and t.tolist() == [0, 1, -100, 2, -100, 1] and p.tolist() == [0, 1, 2, 2, 1, 0]
    assert np.array_equal(cm2.value.numpy(), eval_ref.confusion(p.numpy(), t.numpy(), 3, -100))
    am = AverageMeter()
    am.update(2.0, 3); am.update(4.0, 1)
    assert am.avg == 2.5 and am.val == 4.0 and am.count == 4


def test_product_voxel_parts_and_scatter_mean_on_cpu():
    from amcontrast3d_amd import evaluate
    g, m, parts, _ = setup()
    mine = evaluate.voxel_parts(g["coord"], float(g["voxel"]), rng=np.random.default_rng(3))
    assert len(mine) == len(parts)
    for a, b in zip(mine, parts):
        assert np.array_equal(np.sort(a), np.sort(b))
    rng = np.random.default_rng(0)
    src = rng.standard_normal((500, 13)).astype(np.float32)
    index = rng.integers(0, 120, 500)
    index[:5] = 119
    got = evaluate.scatter_mean(torch.from_numpy(src), torch.from_numpy(index), size=125).numpy()
    want = eval_ref.scatter_mean(src, index, 125)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    assert np.all(got[120:] == 0)


def _amb_expected(g, m):
    amb = m["ambiguity_metrics"]
    return amb, [g[f"amb/cm/{j}"] for j in range(5)]


def bin_edge_free(a, tol=1e-4):
    """points whose bin floor(10 a + 1) cannot flip under a 1e-5 perturbation of a"""
    t = a.astype(np.float64) * 10 + 1
    return np.abs(t - np.round(t)) > tol


def check_ambiguity_outputs(a, mats, ratio, cls, lists, g, m, label):
    """shared by the CPU (oracle) and GPU (product) tests: everything ambiguity_metrics returns against the fixture;
    counts may differ by the points sitting within 1e-5 of a bin edge (none in this fixture, asserted)"""
    amb, ref_mats = _amb_expected(g, m)
    assert np.abs(a - g["amb/a"]).max() <= 1e-5
    moved = int((np.floor(a * np.float32(10) + np.float32(1)) != np.floor(g["amb/a"] * np.float32(10) + np.float32(1))).sum())
    assert moved <= int((~bin_edge_free(g["amb/a"])).sum())
    for got, ref in zip(mats, ref_mats):
        assert np.abs(np.asarray(got) - ref).sum() <= 2 * moved
    if moved == 0:
        assert {float(k): v for k, v in amb["ratio"].items()} == pytest.approx(ratio, rel=1e-12)
        assert {int(k): v for k, v in amb["cls"].items()} == {int(k): v for k, v in cls.items()}
        if lists is not None:
            for key, got in zip(("miou", "macc", "oa"), lists[:3]):
                np.testing.assert_allclose(np.array(got, dtype=np.float64), np.array(amb[key], dtype=np.float64), atol=0.011,
                                           equal_nan=True)
            assert lists[3] == amb["count_per_class"]


def test_ambiguity_metrics_restatement():
    g, m, _, _ = setup()
    r = eval_ref.ambiguity_metrics(g["coord"], g["label"], g["pred"], int(g["nsample"]), m["num_classes"],
                                   m["ignore_index"], 0.04, 0.5)
    lists = [[], [], [], []]
    for mat in r["mats"]:
        with np.errstate(invalid="ignore", divide="ignore"):
            miou, macc, oa, _, _ = eval_ref.get_mious(*eval_ref.tp_union_count(mat))
        lists[0].append(round(miou, 2)); lists[1].append(round(macc, 2)); lists[2].append(round(oa, 2))
        lists[3].append(mat.sum(1).tolist())
    check_ambiguity_outputs(r["a"], r["mats"], r["ratio"], r["cls"], lists, g, m, g["label"])
    amb = m["ambiguity_metrics"]
    assert amb["ratio_low_semi_high"] == [1.0] * 5 and len(amb["count"]) == 5
    assert abs(sum(amb["count"]) - 100) < 0.05


def test_product_voxel_representatives_on_cpu():
    """'nearest_neighbor' test mode: one representative per occupied voxel, every point mapped to the representative
    of its own voxel (main_AA.py:99-107, 666)"""
    from amcontrast3d_amd import evaluate
    g, m, parts, _ = setup()
    coord, v = g["coord"], float(g["voxel"])
    part, expand = evaluate.voxel_representatives(coord, v, rng=np.random.default_rng(1))
    cells = np.floor(coord / v).astype(np.int64)
    assert len(part) == len(parts[0]) == len(np.unique(cells, axis=0))            # one per occupied voxel
    assert len(np.unique(cells[part], axis=0)) == len(part)
    assert expand.shape == (len(coord),) and np.array_equal(cells[part[expand]], cells)
    assert np.array_equal(expand[part], np.arange(len(part)))                       # a representative maps to itself
