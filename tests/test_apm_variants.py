"""The ablation variants of the Ambiguity Prediction Module (APM/attention.py, APM/separation.py of the reference; the
configs list them as alternatives at cfgs/*/AMContrast3D-MM.yaml:40).

Parity unpinned: the reference's classes need CUDA (``.to('cuda')`` inside forward) or cannot be constructed at all
(APM_p: undefined ``self.drop_rate``; APM_p_Graph: undefined ``GCNConv``), so there is no reference run to record.  What is
checked is what reading the reference fixes: the registry names, the state-dict keys (checkpoint compatibility), and the
arithmetic restated with plain torch calls.
"""
import pytest
import torch
import torch.nn as nn


def _activate():
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models import MODELS
    return MODELS


def _fresh_attention_weights(seed, width, dk):
    """the three bias-free maps a fresh Attention(width, dk, width) draws, in its order: query, key, value"""
    torch.manual_seed(seed)
    return [nn.Linear(width, dk, bias=False).weight, nn.Linear(width, dk, bias=False).weight,
            nn.Linear(width, width, bias=False).weight]


def test_all_six_names_are_registered():
    MODELS = _activate()
    for name in ("APM_p", "APM_p_Group", "APM_p_Graph", "APM_pf_ConCate", "APM_pf_CrossAtt", "APM_pp_SelfAtt"):
        assert MODELS.get(name) is not None, name
    # importable from the reference's module paths too (models/backbone/__init__.py:8-10)
    from openpoints.AMContrast3D.APM.attention import APM_pf_CrossAtt, APM_pp_SelfAtt, Attention  # noqa: F401
    from openpoints.AMContrast3D.APM.separation import APM_p, APM_p_Graph, APM_p_Group, KNN  # noqa: F401


def test_cross_attention_variant_keys_and_arithmetic():
    _activate()
    from openpoints.AMContrast3D.APM.attention import APM_pf_CrossAtt
    torch.manual_seed(1)
    apm = APM_pf_CrossAtt(feature_dim=[8, 16, 32, 64], linear_mapping=False).train()
    keys = list(apm.state_dict().keys())
    want = []
    for s in range(4):
        want += [f"layer_{s}.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]
        want += [f"layer_{s}.5.{w}" for w in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    want += [f"ext_{s}.0.{w}" for s in range(4) for w in ("weight", "bias")]
    assert keys == want  # no attention parameters: that layer is rebuilt inside every forward
    B, n, D = 2, 50, 16
    p, f = torch.randn(B, n, 3), torch.randn(B, D, n)
    torch.manual_seed(7)
    got = apm(p, f)
    assert got.shape == (B * n, 1, 1)
    # every point is its own one-token sequence: the softmax over one key is 1, the output is value(f) exactly
    wq, wk, wv = _fresh_attention_weights(7, D, 3)
    rows = f.permute(0, 2, 1).reshape(B * n, D)
    apm2 = APM_pf_CrossAtt(feature_dim=[8, 16, 32, 64], linear_mapping=False).train()
    apm2.load_state_dict(apm.state_dict())
    apm2.layer_1[5].reset_running_stats()
    want_out = apm2.layer_1((rows @ wv.t()).unsqueeze(1))
    torch.testing.assert_close(got, want_out, rtol=1e-6, atol=1e-7)
    # fresh weights per call: a second forward differs, the same seed reproduces
    torch.manual_seed(7)
    again = apm(p, f)
    other = apm(p, f)
    torch.testing.assert_close(again, got, rtol=1e-6, atol=1e-7)
    assert not torch.allclose(other, got)
    assert apm(p, torch.randn(B, 5, n)) is None  # no tower of that width


def test_attention_general_formula():
    """more than one token per sequence: softmax(Q K^T / sqrt(in_dim)) V with in_dim (not dk) under the root"""
    _activate()
    from openpoints.AMContrast3D.APM.attention import Attention
    torch.manual_seed(3)
    att = Attention(6, 3, 4)
    x, y = torch.randn(5, 7 * 6), torch.randn(5, 9 * 6)
    got = att(x, y)
    xs, ys = x.view(5, 7, 6), y.view(5, 9, 6)
    q, k, v = xs @ att.query.weight.t(), ys @ att.key.weight.t(), ys @ att.value.weight.t()
    want = torch.softmax(q @ k.transpose(1, 2) / 6 ** 0.5, dim=-1) @ v
    assert got.shape == (5, 7, 4)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)


def test_self_attention_variant():
    _activate()
    from openpoints.AMContrast3D.APM.attention import APM_pp_SelfAtt
    torch.manual_seed(2)
    apm = APM_pp_SelfAtt().eval()
    assert [k for k in apm.state_dict() if k.endswith("weight")] == [f"layers.{i}.weight" for i in (0, 2, 4, 5)]
    p = torch.randn(2, 40, 3)
    torch.manual_seed(11)
    got = apm(p)
    wv = _fresh_attention_weights(11, 3, 3)[2]
    want = apm.layers((p.reshape(80, 3) @ wv.t()).unsqueeze(1))
    assert got.shape == (80, 1, 1)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)


def test_position_mlp_variant():
    _activate()
    from openpoints.AMContrast3D.APM.separation import APM_p, APM_p_Graph
    apm = APM_p()
    lin = [k for k in apm.state_dict() if k.endswith(".weight") and apm.state_dict()[k].dim() == 2]
    assert lin == [f"layers.{i}.weight" for i in (0, 4, 8, 12, 16, 20)]
    assert [apm.layers[i].p for i in (1, 5, 9, 13, 17)] == [0.2, 0, 0, 0, 0]  # the `dropout` argument (see the module docstring)
    assert isinstance(apm.layers[21], nn.BatchNorm1d) and isinstance(apm.layers[22], nn.Sigmoid) and len(apm.layers) == 23
    out = apm.eval()(torch.randn(3, 10, 3))
    assert out.shape == (30, 1) and float(out.min()) > 0 and float(out.max()) < 1
    with pytest.raises(NotImplementedError, match="GCNConv"):
        APM_p_Graph()


def test_group_variant_keys_and_needs_the_gpu():
    _activate()
    from openpoints.AMContrast3D.APM.separation import APM_p_Group
    apm = APM_p_Group(nsample_k=6)
    sd = apm.state_dict()
    assert [tuple(sd[f"conv.{i}.weight"].shape) for i in (0, 3, 6)] == [(18, 18, 1), (9, 18, 1), (3, 9, 1)]
    assert "conv.0.bias" not in sd and tuple(sd["regressor.weight"].shape) == (1, 3)
    with pytest.raises(Exception):  # the search is the HIP kernel: no CPU path (ops._need_gpu)
        apm(torch.randn(2, 20, 3))


def test_wrapper_routes_the_variants_like_the_reference():
    """base_seg.py:58-86: position-only names call APM(p); the feature variants call APM(p, f) and, with linear_mapping on,
    unpack two results -- which APM_pf_CrossAtt does not return (a (m,1,1) tensor unpacks along its rows and fails)"""
    _activate()
    from amcontrast3d_amd import configs
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig

    class Enc(nn.Module):
        def forward(self, data):
            p = [torch.randn(2, n, 3) for n in (64, 32, 16, 8, 4, 2)]
            f = [torch.randn(2, c, n) for c, n in ((4, 64), (8, 32), (16, 16), (32, 8), (64, 4), (64, 2))]
            return p, f, {}

    class Dec(nn.Module):
        def forward(self, p, f, stage, *rest):
            raise RuntimeError("decoder reached:" + ",".join(sorted(stage)) + f":{tuple(stage['ambiguity'][0].shape)}")

    c = EasyConfig()
    c.update(configs.model_cfg_mm("S", dropout=0, width=8))
    model = build_model_from_cfg(c)
    model.encoder, model.decoder = Enc(), Dec()
    from openpoints.AMContrast3D.APM.attention import APM_pf_CrossAtt, APM_pp_SelfAtt
    model.APM, model.name, model.linear_mapping = APM_pp_SelfAtt(), "APM_pp_SelfAtt", False
    with pytest.raises(RuntimeError, match=r"decoder reached:ambiguity:\(64, 1, 1\)"):
        model({})
    model.APM, model.name = APM_pf_CrossAtt(feature_dim=[8, 16, 32, 64], linear_mapping=False), "APM_pf_CrossAtt"
    with pytest.raises(RuntimeError, match=r"decoder reached:ambiguity,ambiguity_map:\(64, 1, 1\)"):
        model({})
    model.linear_mapping = True
    with pytest.raises(ValueError, match="unpack"):
        model({})


def test_an_apm_module_may_be_the_first_import():
    """separation.py imports openpoints.models.build, whose package registers the APM classes: the cycle must close in
    either order (a fresh interpreter, since this process has long imported both)"""
    import subprocess
    import sys
    code = ("import amcontrast3d_amd; amcontrast3d_amd.activate(); "
            "from openpoints.AMContrast3D.APM.separation import APM_p_Group; "
            "from openpoints.models import MODELS; "
            "assert all(MODELS.get(n) is not None for n in ('APM_p', 'APM_p_Group', 'APM_pf_ConCate', 'APM_pf_CrossAtt'))")
    subprocess.run([sys.executable, "-c", code], check=True, timeout=300)
