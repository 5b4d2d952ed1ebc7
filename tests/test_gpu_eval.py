"""Evaluation path on the MI355X: eval-mode model on the fused inference kernels,
boundary / inner split, voted whole-cloud prediction and confusion matrices, against the fixture recorded from the
reference's own classes (tests/golden/eval_w8_room.npz) and against the oracle (oracle/eval_ref.py).
Indices and counts bit-exact; logits within 1e-4 of their range (BASELINE.json north_star tolerance)."""
import numpy as np
import pytest
import torch

from amcontrast3d_amd import configs
from conftest import load_golden
from test_oracle_eval import NAME, near_tie_free, setup

pytestmark = pytest.mark.gpu


def build_model(m, g, dev):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig
    c = EasyConfig()
    c.update(configs.model_cfg("S", num_classes=m["num_classes"], in_channels=4, dropout=0.5, width=m["width"]))
    model = build_model_from_cfg(c)
    model.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("w/")}, strict=True)
    return model.to(dev).eval()


@pytest.mark.parametrize("shape,pool", [((2, 24, 1000), False), ((3, 16, 77, 32), True), ((1, 8, 50, 20), True),
                                        ((2, 5, 333), False)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_eval_kernels(shape, pool, relu):
    from amcontrast3d_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(shape, generator=g) * 2 + 0.3).to(dev)
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(shape[1]).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(shape[1], generator=g) + 0.5)
        bn.bias.copy_(torch.randn(shape[1], generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(shape[1], generator=g))
        bn.running_var.copy_(torch.rand(shape[1], generator=g) + 0.2)
    bn.eval()
    with torch.no_grad():
        want = bn(x)
        if relu:
            want = torch.relu(want)
        if pool:
            want = want.max(dim=-1)[0]
        got = ops.bn_eval(x, bn, relu, pool)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def test_whole_cloud_matches_reference_run():
    from amcontrast3d_amd import evaluate
    dev = torch.device("cuda:0")
    g, m, parts, boundary = setup()
    model = build_model(m, g, dev)
    label = torch.from_numpy(g["label"]).to(dev)
    nsample = int(g["nsample"])
    r = evaluate.test_cloud_boundary_inner(model, g["coord"], g["feat"], label, parts, m["num_classes"],
                                           m["ignore_index"], nsample)
    scale = max(1.0, float(np.abs(g["voted"]).max()))
    voted = r["logits"].cpu().numpy()
    assert np.abs(voted - g["voted"]).max() <= 1e-4 * scale
    clear = near_tie_free(g["voted"], 2e-4 * scale)
    pred = r["pred"].cpu().numpy()
    assert np.array_equal(pred[clear], g["pred"][clear])
    assert np.abs(r["cm"].value.cpu().numpy() - g["cm/all"]).sum() <= 2 * int((~clear).sum())
    # one sub-cloud per call (the reference's schedule) gives the same logits as the stacked calls
    r1 = evaluate.test_cloud_boundary_inner(model, g["coord"], g["feat"], label, parts, m["num_classes"],
                                            m["ignore_index"], nsample, batch=1)
    assert float((r1["logits"] - r["logits"]).abs().max()) <= 1e-5 * scale
    # the general path (caller-supplied make_input: host-side staging, per-cloud boundary masks) against the default
    def make_input(coord_part, feat_part):
        pos = torch.from_numpy(np.ascontiguousarray(coord_part, dtype=np.float32)).to(dev).unsqueeze(0)
        x = torch.cat([torch.from_numpy(feat_part).to(dev), pos[0, :, 2:3]], 1).t().contiguous().unsqueeze(0)
        return {"pos": pos, "x": x}
    r2 = evaluate.test_cloud_boundary_inner(model, g["coord"], g["feat"], label, parts, m["num_classes"],
                                            m["ignore_index"], nsample, make_input=make_input, batch=2)
    assert float((r2["logits"] - r["logits"]).abs().max()) <= 1e-5 * scale
    for tag in ("cm_b", "cm_i"):
        assert torch.equal(r2[tag].value.sum(1), r[tag].value.sum(1))
    # boundary / inner membership is integer work on labels and neighbour indices: exact
    assert np.array_equal(r["cm_b"].value.sum(1).cpu().numpy(), g["cm/boundary"].sum(1))
    assert np.array_equal(r["cm_i"].value.sum(1).cpu().numpy(), g["cm/inner"].sum(1))
    for j, part in enumerate(parts):
        cp = g["coord"][part]
        cp = torch.from_numpy(np.ascontiguousarray(cp - cp.min(0))).to(dev)
        b = evaluate.boundary_mask(cp, label[torch.from_numpy(part).to(dev)], nsample, m["num_classes"], m["ignore_index"])
        assert np.array_equal(b.cpu().numpy(), boundary[j]), j
    # the confusion matrix / get_mious on the GPU, from the reference's own prediction: exact
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.utils import ConfusionMatrix, get_mious
    cm = ConfusionMatrix(num_classes=m["num_classes"], ignore_index=m["ignore_index"])
    cm.update(torch.from_numpy(g["pred"]).to(dev), label)
    assert np.array_equal(cm.value.cpu().numpy(), g["cm/all"])
    np.testing.assert_allclose(get_mious(cm.tp, cm.union, cm.count)[:3], g["mious/all"], rtol=1e-6)


def test_eval_mode_model_uses_running_statistics_and_matches_torch_modules():
    """the fused inference path against the stored torch modules evaluated one by one (autograd enabled switches the
    fused path off: blocks._eval_bn)"""
    dev = torch.device("cuda:0")
    g, m, parts, _ = setup()
    model = build_model(m, g, dev)
    part = parts[0]
    cp = g["coord"][part]
    cp = cp - cp.min(0)
    pos = torch.from_numpy(np.ascontiguousarray(cp)).to(dev).unsqueeze(0)
    x = torch.cat([torch.from_numpy(g["feat"][part]).to(dev), pos[0, :, 2:3]], 1).t().contiguous().unsqueeze(0)
    with torch.no_grad():
        fused = model({"pos": pos, "x": x})[0]
    plain = model({"pos": pos, "x": x})[0].detach()  # grad mode on: nn.BatchNorm modules in eval mode
    scale = max(1.0, float(plain.abs().max()))
    assert float((fused - plain).abs().max()) <= 1e-4 * scale
    assert float((fused[0].cpu() - torch.from_numpy(g["logits/0"])).abs().max()) <= 1e-4 * scale
    for mod in model.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            assert int(mod.num_batches_tracked) == 2  # untouched by the evaluation


def test_validate_boundary_inner_against_oracle():
    from amcontrast3d_amd import evaluate, synthetic
    from oracle import eval_ref
    dev = torch.device("cuda:0")
    g, m, _, _ = setup()
    model = build_model(m, g, dev)
    nsample, ncls = int(g["nsample"]), m["num_classes"]
    batches, preds = [], []
    for k in range(2):
        nb = synthetic.make_batch(1, 3000, first_id=700 + k, num_classes=ncls)
        noise = np.random.default_rng(k)
        flip = noise.random(3000) < 0.05
        nb["y"][0, flip] = noise.integers(0, ncls, int(flip.sum()))
        batches.append({k_: torch.from_numpy(v).to(dev) for k_, v in nb.items()})
    out = evaluate.validate_boundary_inner(model, batches, ncls, m["ignore_index"], nsample)
    assert len(out) == 15
    cm = np.zeros((ncls, ncls), np.int64); cm_b = cm.copy(); cm_i = cm.copy()
    with torch.no_grad():
        for d in batches:
            pred = model(d)[0].argmax(1)[0].cpu().numpy()
            y = d["y"][0].cpu().numpy()
            b = eval_ref.boundary_mask(d["pos"][0].cpu(), y, nsample, ncls, m["ignore_index"])
            cm += eval_ref.confusion(pred, y, ncls); cm_b += eval_ref.confusion(pred[b], y[b], ncls)
            cm_i += eval_ref.confusion(pred[~b], y[~b], ncls)
    assert cm_i.sum() > 0 and cm_b.sum() > 0
    for j, mat in enumerate((cm, cm_b, cm_i)):
        want = eval_ref.get_mious(*eval_ref.tp_union_count(mat))
        np.testing.assert_allclose(out[5 * j:5 * j + 3], want[:3], rtol=2e-6)
        np.testing.assert_allclose(out[5 * j + 3], want[3], rtol=2e-6)


def test_ambiguity_metrics_matches_reference_run():
    """openpoints.AMContrast3D.metrics.ambiguity_metrics on the GPU against what the reference's own function returned
    for the same cloud, labels and prediction (fixture meta 'ambiguity_metrics', amb/*)"""
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.AMContrast3D.metrics import ambiguity_metrics, posmask_searching
    from openpoints.utils import ConfusionMatrix
    from test_oracle_eval import check_ambiguity_outputs
    dev = torch.device("cuda:0")
    g, m, _, _ = setup()
    p = torch.from_numpy(g["coord"]).to(dev)
    label, pred = torch.from_numpy(g["label"]).to(dev), torch.from_numpy(g["pred"]).to(dev)
    nsample = int(g["nsample"])
    posmask, nidx = posmask_searching(p, label, nsample, m["num_classes"], m["ignore_index"])
    cms = [ConfusionMatrix(num_classes=m["num_classes"], ignore_index=m["ignore_index"]) for _ in range(5)]
    a, ratio, count, ratio_lsh, cls, l_miou, l_macc, l_oa, l_count = ambiguity_metrics(
        p, label, pred, posmask, nsample, nidx, "Method2", 0.04, False, *cms, 0.5)
    check_ambiguity_outputs(a.cpu().numpy(), [c.value.cpu().numpy() for c in cms], ratio, cls,
                            [l_miou, l_macc, l_oa, l_count], g, m, g["label"])
    amb = m["ambiguity_metrics"]
    assert ratio_lsh == amb["ratio_low_semi_high"]
    np.testing.assert_allclose(list(count), amb["count"], atol=0.011)


def test_nearest_neighbour_mode():
    from amcontrast3d_amd import evaluate
    dev = torch.device("cuda:0")
    g, m, _, _ = setup()
    model = build_model(m, g, dev)
    label = torch.from_numpy(g["label"]).to(dev)
    part, expand = evaluate.voxel_representatives(g["coord"], float(g["voxel"]), rng=np.random.default_rng(1))
    r = evaluate.test_cloud_boundary_inner(model, g["coord"], g["feat"], label, [part], m["num_classes"], m["ignore_index"],
                                           int(g["nsample"]), expand=expand)
    single = evaluate.test_cloud_boundary_inner(model, g["coord"][part], g["feat"][part], label[torch.from_numpy(part).to(dev)],
                                                [np.arange(len(part))], m["num_classes"], m["ignore_index"], int(g["nsample"]))
    assert r["pred"].shape == label.shape
    assert torch.equal(r["logits"], single["logits"][torch.from_numpy(expand).to(dev)])
    assert int(r["cm"].total) == len(label) and int(r["cm_b"].total + r["cm_i"].total) == len(part)
