"""GPU side of tests/test_apm_variants.py: the grouped variant's neighbourhoods against a brute-force search, and every
constructible APM variant inside the AMContrast3D++ model (forward, three-term loss, backward)."""
import pytest
import torch

from amcontrast3d_amd import configs

pytestmark = pytest.mark.gpu


def easy(d):
    from openpoints.utils import EasyConfig
    c = EasyConfig()
    c.update(d)
    return c


def test_group_variant_against_brute_force_neighbourhoods():
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.AMContrast3D.APM.separation import KNN, APM_p_Group
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, n, k = 3, 700, 12
    p = torch.rand(B, n, 3, device=dev)
    rows = p.reshape(B * n, 3)
    idx, nbr = KNN(rows, k)
    d = torch.cdist(rows.double(), rows.double())
    want = d.topk(k, dim=1, largest=False).indices[:, 1:]  # all clouds of the batch are ONE point list there (separation.py:64)
    assert idx.shape == (B * n, k - 1) and torch.equal(idx.long(), want)
    assert torch.equal(nbr, rows[want])
    apm = APM_p_Group(nsample_k=k).to(dev).train()
    got = apm(p)
    rel = torch.cat([rows.unsqueeze(1), (rows.unsqueeze(1) - rows[want]).abs()], dim=1).reshape(B, n, 3 * k).transpose(1, 2)
    ref = APM_p_Group(nsample_k=k).to(dev).train()
    ref.load_state_dict(apm.state_dict())
    for m in ref.conv:
        if hasattr(m, "reset_running_stats"):
            m.reset_running_stats()
    out = ref.regressor(ref.conv(rel).transpose(1, 2))
    out = torch.softmax(out, dim=0).reshape(B * n, 1)  # across the clouds: every column of B values sums to one
    assert got.shape == (B * n, 1)
    torch.testing.assert_close(got, out, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(got.view(B, n).sum(0), torch.ones(n, device=dev), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["APM_p", "APM_p_Group", "APM_pp_SelfAtt", "APM_pf_CrossAtt"])
def test_variants_inside_the_model(name):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import synthetic
    from openpoints.loss import build_criterion_from_cfg
    from openpoints.models import build_model_from_cfg
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = {"channel": [64, 32, 16, 8, 4, 2], "dropout": [0.2, 0, 0, 0, 0, 0]} if name in ("APM_p", "APM_p_Group") else {}
    cfg = configs.model_cfg_mm("S", dropout=0, width=8, threshold=0.5, NAME=name)
    cfg["APM_args"].update(kw)  # (the six-entry lists are separation.py's defaults; the yaml block carries ConCate's five)
    model = build_model_from_cfg(easy(cfg)).to(dev).train()
    criterion = build_criterion_from_cfg(easy(configs.criterion_cfg_mm())).to(dev)
    aa = easy(configs.ambiguity_args_mm("s3dis"))
    data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(2, 2048, first_id=9).items()}
    logits, stage, refine = model(dict(data))
    assert logits.shape == (2, 13, 2048) and len(stage["ambiguity"]) == 4
    assert [a.numel() for a in stage["ambiguity"]] == [2 * 2048 // 4 ** i for i in range(4)]  # p[1..4]
    assert ("ambiguity_map" in stage) == (name == "APM_pf_CrossAtt")
    seg, ce, contrast, reg = criterion(logits, data["y"], stage, 13, None, aa)
    (seg + reg).backward()
    assert all(torch.isfinite(t).all() for t in (logits, seg, ce, contrast, reg)) and 0.0 <= refine <= 100.0
    grads = [p.grad for p in model.APM.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads) and any(float(g.abs().max()) > 0 for g in grads)
