"""Fused residual branch of a strided SetAbstraction block (csrc/sa_res.hip) against the torch operators it replaces
(pointnext_AA.py:157-168: torch.gather at the FPS picks -> Conv1d with bias -> add -> ReLU), forward and backward, with an
fp64 evaluation as the arbiter; and the model-level switch (same logits and gradients with and without the fused branch)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference(y, f, idx, w, b, gout, dtype):
    leaves = [t.to(dtype).requires_grad_(True) for t in (y, f, w, b)]
    fi = torch.gather(leaves[1], -1, idx.long().unsqueeze(1).expand(-1, f.shape[1], -1))
    out = F.relu(leaves[0] + F.conv1d(fi, leaves[2], leaves[3]))
    out.backward(gout.to(dtype))
    return [out.detach()] + [t.grad for t in leaves]


# the four stages of PointNeXt-S at 24000 points per cloud (M = 6000, 1500, 375, 93), ragged channel counts, tiny clouds
@pytest.mark.parametrize("B,Cin,Cout,N,M", [(2, 32, 64, 24000, 6000), (2, 64, 128, 6000, 1500), (3, 128, 256, 1500, 375),
                                             (8, 256, 512, 375, 93), (2, 24, 40, 1000, 333), (1, 3, 5, 70, 9),
                                             (2, 70, 130, 300, 65)])
def test_sa_residual_matches_torch(B, Cin, Cout, N, M):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(Cin * 13 + Cout)
    y = torch.randn(B, Cout, M, generator=g).to(DEV)
    f = torch.randn(B, Cin, N, generator=g).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).to(torch.int32).to(DEV)  # distinct per cloud
    w = (torch.randn(Cout, Cin, 1, generator=g) * (1.0 / Cin ** 0.5)).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    gout = torch.randn(B, Cout, M, generator=g).to(DEV)
    leaves = [t.clone().requires_grad_(True) for t in (y, f, w, b)]
    out = ops.sa_residual(leaves[0], leaves[1], idx, leaves[2], leaves[3])
    out.backward(gout)
    got = [out.detach()] + [t.grad for t in leaves]
    r64 = reference(y, f, idx, w, b, gout, torch.float64)
    r32 = reference(y, f, idx, w, b, gout, torch.float32)
    for name, a, t64, t32 in zip(("out", "dy", "df", "dw", "db"), got, r64, r32):
        scale = float(t64.abs().max()) + 1e-30
        err = float((a.double() - t64).abs().max()) / scale
        err32 = float((t32.double() - t64).abs().max()) / scale
        assert err <= max(4 * err32, 2e-6), (name, err, err32)
    # ReLU mask: exactly the elements torch keeps (pre-activations within rounding of zero aside)
    flips = int(((got[0] > 0) != (r32[0] > 0)).sum())
    assert flips <= 1 + got[0].numel() // 100000, flips
    # untouched columns of df are exactly zero
    hit = torch.zeros(B, N, dtype=torch.bool, device=DEV)
    hit.scatter_(1, idx.long(), True)
    assert float(got[2].abs().amax(1)[~hit].max() if (~hit).any() else 0.0) == 0.0


def test_sa_residual_sums_over_repeated_picks():
    """A cloud with fewer distinct points than picks (crop_pc pads small rooms by repetition, dataset/data_util.py:161-167) makes
    FPS return an index more than once; torch.gather's backward sums the gradient over repeated indices and the fused branch
    must too (plain stores kept one racing writer's value)."""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(77)
    B, Cin, Cout, N, M = 2, 32, 64, 600, 300
    y = torch.randn(B, Cout, M, generator=g).to(DEV)
    f = torch.randn(B, Cin, N, generator=g).to(DEV)
    idx = torch.randint(0, 40, (B, M), generator=g).to(torch.int32).to(DEV)  # 300 picks among 40 points: every one repeated
    w = (torch.randn(Cout, Cin, 1, generator=g) * (1.0 / Cin ** 0.5)).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    gout = torch.randn(B, Cout, M, generator=g).to(DEV)
    leaves = [t.clone().requires_grad_(True) for t in (y, f, w, b)]
    out = ops.sa_residual(leaves[0], leaves[1], idx, leaves[2], leaves[3])
    out.backward(gout)
    r64 = reference(y, f, idx, w, b, gout, torch.float64)
    for name, a, t64 in zip(("out", "dy", "df", "dw", "db"), [out.detach()] + [t.grad for t in leaves], r64):
        scale = float(t64.abs().max()) + 1e-30
        assert float((a.double() - t64).abs().max()) / scale <= 1e-5, name


def test_the_plans_duplicate_flag_selects_plain_stores_only_for_distinct_picks():
    """ops.index_duplicates (part of the sampling plan) tells the backward whether some cloud's picks repeat an index: 0 ->
    the scatter of df is plain stores (bit for bit the adds into the zero-filled df: one add per address), 1 -> it adds, and the
    repeated picks sum as torch.gather's backward does."""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, N, M = 3, 32, 64, 2000, 500
    y = torch.randn(B, Cout, M, generator=g).to(DEV)
    f = torch.randn(B, Cin, N, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 1, generator=g) * (1.0 / Cin ** 0.5)).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.3).to(DEV)
    gout = torch.randn(B, Cout, M, generator=g).to(DEV)
    distinct = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).to(torch.int32).to(DEV)
    repeated = distinct.clone()
    repeated[1, 7] = repeated[1, 400]  # one repeat in one cloud
    outside = distinct.clone()
    outside[2, 0] = N  # an index outside the cloud counts as "not distinct" (the adds are right for any picks)
    assert int(ops.index_duplicates(distinct, N)) == 0 and int(ops.index_duplicates(repeated, N)) == 1
    assert int(ops.index_duplicates(outside, N)) == 1

    def grads(idx, flag):
        leaves = [t.clone().requires_grad_(True) for t in (y, f, w, b)]
        out = ops.sa_residual(leaves[0], leaves[1], idx, leaves[2], leaves[3], flag)
        out.backward(gout)
        return [out.detach()] + [t.grad for t in leaves]

    for a, c in zip(grads(distinct, None), grads(distinct, ops.index_duplicates(distinct, N))):
        assert torch.equal(a, c)
    r64 = reference(y, f, repeated, w, b, gout, torch.float64)
    for name, a, t64 in zip(("out", "dy", "df", "dw", "db"), grads(repeated, ops.index_duplicates(repeated, N)), r64):
        assert float((a.double() - t64).abs().max()) / (float(t64.abs().max()) + 1e-30) <= 1e-5, name


def test_sa_residual_is_deterministic_and_graph_safe():
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, N, M = 4, 64, 128, 6000, 1500
    y = torch.randn(B, Cout, M, generator=g).to(DEV).requires_grad_(True)
    f = torch.randn(B, Cin, N, generator=g).to(DEV).requires_grad_(True)
    idx = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).to(torch.int32).to(DEV)
    w = torch.randn(Cout, Cin, 1, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(Cout, generator=g).to(DEV).requires_grad_(True)
    gout = torch.randn(B, Cout, M, generator=g).to(DEV)

    def run():
        for t in (y, f, w, b):
            t.grad = None
        out = ops.sa_residual(y, f, idx, w, b)
        out.backward(gout)
        return [out.detach().clone()] + [t.grad.clone() for t in (y, f, w, b)]

    first = run()
    for _ in range(3):
        for a, c in zip(first, run()):
            assert torch.equal(a, c)
    # replayed from a hipGraph: bit-identical to the eager launches
    static = [t.detach().clone().requires_grad_(True) for t in (y, f, w, b)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            o = ops.sa_residual(static[0], static[1], idx, static[2], static[3])
            grads = torch.autograd.grad(o, static, gout)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o = ops.sa_residual(static[0], static[1], idx, static[2], static[3])
        grads = torch.autograd.grad(o, static, gout)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a, c in zip(first, [o] + list(grads)):
        assert torch.equal(a, c)


def test_model_with_and_without_fused_residual(monkeypatch):
    """PointNeXt-S step (sa_use_res): fused residual branch vs the torch operators -- logits to 1e-5 of their range,
    every parameter gradient to 2e-3 norm-wise (max-pool routing near ties flips between any two fp32 evaluations)"""
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import configs, synthetic, timing
    from openpoints.loss import build_criterion_from_cfg
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig
    torch.manual_seed(0)
    c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0))
    model = build_model_from_cfg(c).to(DEV).train()
    cc = EasyConfig(); cc.update(configs.criterion_cfg())
    crit = build_criterion_from_cfg(cc).to(DEV)
    aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
    data = {k: torch.from_numpy(v).to(DEV) for k, v in synthetic.make_batch(2, 4096, first_id=3).items()}
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def step():
        model.load_state_dict(state)
        for p in model.parameters():
            p.grad = None
        with timing.count_calls() as calls:
            logits, stage = model(dict(data))
            loss = crit(logits, data["y"], stage, 13, None, aa)
            loss.backward()
        return logits.detach().clone(), float(loss), {k: p.grad.clone() for k, p in model.named_parameters()}, dict(calls)

    l1, loss1, g1, calls1 = step()
    monkeypatch.setenv("AMC3D_NO_SA_RESIDUAL", "1")
    l0, loss0, g0, calls0 = step()
    assert calls1.get("sa_residual_forward") == 4 and calls1.get("sa_residual_backward") == 4, calls1
    assert "sa_residual_forward" not in calls0
    rng = float(l0.max() - l0.min())
    assert float((l1 - l0).abs().max()) <= 1e-5 * rng
    assert abs(loss1 - loss0) <= 1e-5 * max(1.0, abs(loss0))
    for k in g0:
        assert float((g1[k] - g0[k]).norm()) <= 2e-3 * float(g0[k].norm()) + 1e-7, k
