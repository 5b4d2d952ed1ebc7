"""Per-layer parity at the real widths and position counts of BASELINE's configurations.

End to end, 17-58 batch-statistics BatchNorms amplify fp32 rounding until two correct evaluations of PointNeXt-L / -XL
differ by more than north_star's 1e-4 (tests/test_gpu_fullsize.py measures that against fp64).  A single layer is
well conditioned, so here the tolerance IS 1e-4 of the tensor's range, on the output and on every input / weight
gradient, with the ORACLE's layer input fed to both sides:

    oracle/model_ref.py  set_abstraction / inv_res_mlp / feature_propagation   (the layer as the reference writes it:
                         pointnext_AA.py:139-170, 57-63 + 296-307, 210-226)  on oracle/pointops_ref.c
    product              SetAbstraction / InvResMLP / FeaturePropogation of amcontrast3d_amd.openpoints on the C-ABI kernels

Every fused layer kind the bench's configurations dispatch is covered at L and XL channel counts and full position
counts, with the dispatch asserted (timing.count_calls):

    LocalAggregationFused   single-conv SetAbstraction of L / XL and the LocalAggregation of every InvResMLP (csrc/lagg.hip)
    bn_residual             relu(bn(x) + identity) at the end of InvResMLP (csrc/bn.hip)
    GroupedConvBN + SATailActivated + sa_residual    PointNeXt-S' two-layer SetAbstraction with skip conv
    GroupedConvBN + gm_gemm + bn_max + sa_residual   the same at SA2-4 (the recomputing tail does not pay there)
    FeaturePropagation first block   conv before the 3-NN interpolation (three_interpolate_add) + BN + ReLU

Forward: against the oracle's own forward (its max-pools return their own maxima).  The product's arg-max picks are
only compared: where one differs from torch.max's, the oracle's maximum may exceed the value at the product's pick by
at most 1e-5 of the tensor's range (identical inputs: a near-tie within rounding), and at most 1e-4 of the picks may
differ.  Gradients: against the oracle routed through the product's picks (one element per (b,c,m) receives the
gradient; a near-tie flip would otherwise move O(1) gradient mass between two neighbours).  ReLU masks are the other
discontinuity: the layer's biases are first moved off every pre-activation within rounding of zero (_open_relu_margins;
without it 1-4 elements per case flip and take whole rows of the weight gradients with them -- measured).
Coordinates: the cloud of the stage in question, reached by FPS from a synthetic S3DIS-/ScanNet-shaped batch (the
sampler itself is bit-exact: tests/test_gpu_ops.py).  No fixture of the reference covers these sizes (CUDA-only natives):
unpinned by the reference, pinned by the oracle that the small fixtures pin.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4  # north_star: logits/loss within 1e-4 fp32 -- here of every compared tensor's range


def _easy(**kw):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.utils import EasyConfig
    c = EasyConfig()
    c.update(kw)
    return c


def _cloud(B, N0, level, voxel):
    """coordinates of encoder stage `level` (N0 / 4**level points per cloud) of a synthetic batch, on the CPU"""
    from amcontrast3d_amd import ops, synthetic
    p = torch.from_numpy(synthetic.make_batch(B, N0, first_id=700 + level, voxel_size=voxel)["pos"]).to(DEV)
    for _ in range(level):
        idx = ops.furthest_point_sample(p, p.shape[1] // 4).long()
        p = torch.gather(p, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    return p.cpu().contiguous()


def _randomise(mod, seed):
    """default init leaves every BatchNorm at gamma 1 / beta 0: draw them (and signs) so the affine part is exercised"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.weight[::7] *= -1
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)


def _threads():
    from oracle import pointops_ref
    n = min(len(os.sched_getaffinity(0)), 16)
    pointops_ref.set_threads(n)
    torch.set_num_threads(n)


def _check(name, got, want, floor=0.0):
    rng = max(float(want.abs().max()), floor, 1e-30)
    d = (got.detach().cpu().float() - want.float()).abs()
    err = float(d.max())
    if err > TOL * rng:  # where: how many distinct indices along every axis hold an element over the tolerance
        bad = (d > TOL * rng).nonzero()
        where = [f"axis {a}: {bad[:, a].unique().numel()} distinct (first {bad[:, a].unique()[:6].tolist()})" for a in range(bad.shape[1])]
        raise AssertionError((name, err, rng, err / rng, f"{bad.shape[0]} of {d.numel()} elements over", where))
    return err / rng


def _open_relu_margins(prefix, mod, oracle_forward, inputs):
    """Move the per-channel additive parameter in front of every ReLU (BatchNorm bias / skip-conv bias) by a few 1e-5 until
    no pre-activation of the ORACLE's fp32 forward lies within `delta` of zero.  At an element within rounding of zero two
    correct fp32 evaluations disagree on the ReLU mask and every gradient downstream differs by a whole term (one term of a
    weight gradient summed over 1e4-1e5 positions is 1e-3 of its range): a property of the layer, not of an
    implementation -- so the test's parameters are chosen off those discontinuities, and then the tolerance can be 1e-4.
    delta = 2e-5 where a channel has <= 1e4 values, down to 2e-6 for the dense (B,C,M,32) activation of PointNeXt-S' first
    MLP layer (product and oracle differ by <= 5e-7 there)."""
    from oracle import model_ref
    params = {f"{prefix}.{k}": p for k, p in mod.named_parameters()}
    for it in range(8):
        sd = {f"{prefix}.{k}": v.detach().cpu().clone() for k, v in mod.state_dict().items()}
        moved = []

        def probe(key, v):
            C = v.shape[1]
            flat = v.transpose(0, 1).reshape(C, -1)
            delta = min(2e-5, max(2e-6, 0.3 / flat.shape[1]))
            bad = (flat.abs() < delta).any(1)
            if not bool(bad.any()):
                return
            rows = flat[bad]
            shift = torch.zeros(rows.shape[0])
            todo = torch.ones(rows.shape[0], dtype=torch.bool)
            for j in range(1, 2000):
                s = ((j + 1) // 2) * (1 if j % 2 else -1) * 2.5 * delta
                left = todo.nonzero()[:, 0]
                ok = ~((rows[left] + s).abs() < delta).any(1)
                shift[left[ok]] = s
                todo[left[ok]] = False
                if not bool(todo.any()):
                    break
            assert not bool(todo.any()), ("no gap found", key)
            with torch.no_grad():
                params[key][bad.nonzero()[:, 0].to(DEV)] += shift.to(DEV)
            moved.append((key, int(bad.sum())))

        model_ref.RELU_PROBE = probe
        try:
            with torch.no_grad():
                oracle_forward(sd, None, **inputs)
        finally:
            model_ref.RELU_PROBE = None
        if not moved:
            return it
    raise AssertionError(("the ReLU margins did not settle", moved))


def _run(prefix, mod, gpu_forward, oracle_forward, inputs, npools, expect_calls):
    """inputs: dict name -> CPU tensor that receives a gradient.  gpu_forward(mod, **gpu leaves) / oracle_forward(sd, pool,
    **cpu leaves) -> output tensor."""
    from amcontrast3d_amd import ops, timing
    from oracle import model_ref
    _threads()
    passes = _open_relu_margins(prefix, mod, oracle_forward, inputs)
    sd = {f"{prefix}.{k}": v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    leaves = {k: v.to(DEV).requires_grad_(True) for k, v in inputs.items()}
    log = {}
    ops.pool_log(log)
    try:
        with timing.count_calls() as calls:
            out = gpu_forward(mod, **leaves)
            go = torch.randn(out.shape, generator=torch.Generator().manual_seed(99))
            out.backward(go.to(DEV))
            torch.cuda.synchronize()
            calls = dict(calls)
    finally:
        ops.pool_log(None)
    for k, n in expect_calls.items():
        assert calls.get(k, 0) == n, (k, calls)
    assert len(log) == npools, (len(log), npools)
    picks = {i: a.cpu() for i, a in log.items()}
    # forward against the oracle's own forward
    watch = model_ref.PoolRouting(compare=picks)
    with torch.no_grad():
        own = oracle_forward(sd, watch, **inputs)
    worst = {"out": _check("output", out, own)}
    if npools:
        flips = sum(watch.flips.values())
        gap = max([float(g.max()) for g in watch.gaps.values()] or [0.0])
        assert flips <= 1e-4 * watch.total + 1, ("max-pool picks differing from torch.max's", flips, watch.total)
        assert gap <= 1e-5, ("a differing max-pool pick is not a near-tie", gap)
        worst["flips"], worst["gap"] = flips, gap
    # gradients against the routed oracle
    leaf_sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    cin = {k: v.clone().requires_grad_(True) for k, v in inputs.items()}
    ref = oracle_forward(leaf_sd, model_ref.PoolRouting(picks), **cin)
    ref.backward(go)
    failures = []
    named = dict(mod.named_parameters())
    for k, g_got, g_want in ([(k, leaves[k].grad, cin[k].grad) for k in inputs]
                             + [(k, p.grad, leaf_sd[f"{prefix}.{k}"].grad) for k, p in named.items()]):
        if g_want is None:
            assert g_got is None or float(g_got.abs().max()) == 0.0, k
            continue
        # a BatchNorm's d(beta) = sum dz and d(gamma) = sum dz * xhat are sums of the same terms; where another BatchNorm
        # follows (InvResMLP's LocalAggregation -> pwconv) d(beta) cancels to ~0 analytically and is judged on d(gamma)'s scale
        floor = 0.0
        if k.endswith(".bias") and k[:-5] + ".weight" in named and named[k[:-5] + ".weight"].dim() == 1:
            floor = float(leaf_sd[f"{prefix}.{k[:-5]}.weight"].grad.abs().max())
        try:
            worst["d" + k] = _check("d" + k, g_got, g_want, floor)
        except AssertionError as e:
            failures.append(e.args[0])
            print("FAILURE", e.args[0])
    assert not failures, [f[0] for f in failures]
    print(f"[{prefix}] ReLU margins opened in {passes} passes; worst relative errors: " + ", ".join(f"{k} {v:.1e}" if isinstance(v, float) else f"{k} {v}" for k, v in worst.items()))


# ---------------------------------------------------------------------------------------------------------------------
# SetAbstraction, single conv layer (PointNeXt-B / -L / -XL): LocalAggregationFused
#   (tag, B, N0, level of the SUPPORT cloud, Cin, Cout, radius, voxel)
SA1 = [
    ("L-stage1", 8, 24000, 0, 32, 64, 0.1, 0.04),
    ("L-stage3", 8, 24000, 2, 128, 256, 0.4, 0.04),
    ("L-stage4", 8, 24000, 3, 256, 512, 0.8, 0.04),
    ("XL-stage1-cfg4", 2, 64000, 0, 64, 128, 0.1, 0.02),
    ("XL-stage2-cfg4", 2, 64000, 1, 128, 256, 0.2, 0.02),
    ("XL-stage4-cfg4", 2, 64000, 3, 512, 1024, 0.8, 0.02),
]


@pytest.mark.parametrize("tag,B,N0,level,cin,cout,radius,voxel", SA1)
def test_set_abstraction_single_layer(tag, B, N0, level, cin, cout, radius, voxel):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models.backbone.pointnext_blocks import SetAbstraction
    from oracle import model_ref
    torch.manual_seed(1)
    p = _cloud(B, N0, level, voxel)
    f = torch.randn(B, cin, p.shape[1], generator=torch.Generator().manual_seed(2))
    mod = SetAbstraction(cin, cout, 1, 4, group_args=_easy(NAME='ballquery', radius=radius, nsample=32, normalize_dp=True),
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                         feature_type='dp_fj', use_res=False).to(DEV).train()
    _randomise(mod, 3)
    pg = p.to(DEV)
    _run("blk", mod, lambda m, f: m([pg, f])[1],
         lambda sd, pool, f: model_ref.set_abstraction(sd, "blk", p, f, 4, radius, 32, 1, False, True, True, pool)[1],
         {"f": f}, 1, {"local_aggregation_forward": 1, "local_aggregation_backward": 1})


# ---------------------------------------------------------------------------------------------------------------------
# InvResMLP: LocalAggregationFused (self query) + pointwise C -> 4C -> C + relu(bn(x) + identity)
INV = [
    ("L-stage1", 8, 24000, 1, 64, 0.2, 0.04),
    ("L-stage2", 8, 24000, 2, 128, 0.4, 0.04),
    ("L-stage4", 8, 24000, 4, 512, 1.6, 0.04),
    ("XL-stage1-cfg4", 2, 64000, 1, 128, 0.2, 0.02),
    ("XL-stage2-cfg4", 2, 64000, 2, 256, 0.4, 0.02),
    ("XL-stage4-cfg4", 2, 64000, 4, 1024, 1.6, 0.02),
]


@pytest.mark.parametrize("tag,B,N0,level,C,radius,voxel", INV)
def test_inv_res_mlp(tag, B, N0, level, C, radius, voxel):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models.backbone.pointnext_blocks import InvResMLP
    from oracle import model_ref
    torch.manual_seed(4)
    p = _cloud(B, N0, level, voxel)
    f = torch.randn(B, C, p.shape[1], generator=torch.Generator().manual_seed(5))
    mod = InvResMLP(C, norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, aggr_args={'feature_type': 'dp_fj', 'reduction': 'max'},
                    group_args=_easy(NAME='ballquery', radius=radius, nsample=32, normalize_dp=True),
                    conv_args={'order': 'conv-norm-act'}, expansion=4, use_res=True).to(DEV).train()
    _randomise(mod, 6)
    pg = p.to(DEV)
    _run("blk", mod, lambda m, f: m([pg, f])[1],
         lambda sd, pool, f: model_ref.inv_res_mlp(sd, "blk", p, f, radius, 32, True, True, pool),
         {"f": f}, 1, {"local_aggregation_forward": 1, "local_aggregation_backward": 1, "bn_residual_forward": 1})


# ---------------------------------------------------------------------------------------------------------------------
# SetAbstraction of PointNeXt-S: two conv layers + skip conv of the sampled features + ReLU
SA2 = [
    ("S-stage1", 8, 24000, 0, 32, 64, 0.1, {"grouped_conv_bn_forward": 1, "sa_tail_forward": 1, "sa_residual_forward": 1}),
    ("S-stage2", 8, 24000, 1, 64, 128, 0.2, {"grouped_conv_bn_forward": 1, "sa_tail_forward": 0, "bn_max_forward": 1, "sa_residual_forward": 1}),
    ("S-stage3", 8, 24000, 2, 128, 256, 0.4, {"grouped_conv_bn_forward": 1, "bn_max_forward": 1, "sa_residual_forward": 1}),
    ("S-stage4", 8, 24000, 3, 256, 512, 0.8, {"grouped_conv_bn_forward": 1, "bn_max_forward": 1, "sa_residual_forward": 1}),
]


@pytest.mark.parametrize("tag,B,N0,level,cin,cout,radius,calls", SA2)
def test_set_abstraction_two_layers_with_skip(tag, B, N0, level, cin, cout, radius, calls):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.models.backbone.pointnext_blocks import SetAbstraction
    from oracle import model_ref
    torch.manual_seed(7)
    p = _cloud(B, N0, level, 0.04)
    f = torch.randn(B, cin, p.shape[1], generator=torch.Generator().manual_seed(8))
    mod = SetAbstraction(cin, cout, 2, 4, group_args=_easy(NAME='ballquery', radius=radius, nsample=32, normalize_dp=True),
                         norm_args={'norm': 'bn'}, act_args={'act': 'relu'}, conv_args={'order': 'conv-norm-act'},
                         feature_type='dp_fj', use_res=True).to(DEV).train()
    _randomise(mod, 9)
    pg = p.to(DEV)
    _run("blk", mod, lambda m, f: m([pg, f])[1],
         lambda sd, pool, f: model_ref.set_abstraction(sd, "blk", p, f, 4, radius, 32, 2, True, True, True, pool)[1],
         {"f": f}, 1, calls)


# ---------------------------------------------------------------------------------------------------------------------
# FeaturePropagation: 3-NN interpolation + concat + [Conv1d, BN, ReLU] x 2 -- first conv before the interpolation
#   (tag, B, N0, level of the FINE cloud, C_skip, C_coarse, C_out, voxel)
FP = [
    ("S/L-finest", 8, 24000, 0, 32, 64, 32, 0.04),
    ("L-level2", 8, 24000, 2, 128, 256, 128, 0.04),
    ("L-coarsest", 8, 24000, 3, 256, 512, 256, 0.04),
    ("XL-finest-cfg4", 2, 64000, 0, 64, 128, 64, 0.02),
    ("XL-coarsest-cfg4", 2, 64000, 3, 512, 1024, 512, 0.02),
]


@pytest.mark.parametrize("tag,B,N0,level,cskip,ccoarse,cout,voxel", FP)
def test_feature_propagation(tag, B, N0, level, cskip, ccoarse, cout, voxel):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import ops
    from openpoints.models.backbone.pointnext_blocks import FeaturePropogation
    from oracle import model_ref
    torch.manual_seed(10)
    p1 = _cloud(B, N0, level, voxel)
    idx = ops.furthest_point_sample(p1.to(DEV), p1.shape[1] // 4).long()
    p2 = torch.gather(p1.to(DEV), 1, idx.unsqueeze(-1).expand(-1, -1, 3)).cpu().contiguous()
    g = torch.Generator().manual_seed(11)
    f1 = torch.randn(B, cskip, p1.shape[1], generator=g)
    f2 = torch.randn(B, ccoarse, p2.shape[1], generator=g)
    mod = FeaturePropogation([cskip + ccoarse, cout, cout]).to(DEV).train()
    _randomise(mod, 12)
    p1g, p2g = p1.to(DEV), p2.to(DEV)
    _run("dec", mod, lambda m, f1, f2: m([p1g, f1], [p2g, f2]),
         lambda sd, pool, f1, f2: model_ref.feature_propagation(sd, "dec", p1, f1, p2, f2, True),
         {"f1": f1, "f2": f2}, 0, {"three_interpolate": 1, "three_interpolate_grad": 1})
