"""Fused adaptive-margin contrast kernels against the oracle's torch-CPU restatement and the
reference-run fixture (tests/golden/ops_small.npz amb/*)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ambiguity_matches_reference_run():
    from amcontrast3d_amd import ops
    g = load_golden("ops_small")
    p = torch.from_numpy(g["amb/p"]).to(DEV)
    lab = torch.from_numpy(g["amb/label"]).to(DEV).to(torch.int32)
    nidx = torch.from_numpy(g["amb/nidx"]).to(DEV)
    posmask = ops.posmask_from_labels(lab, nidx)
    want_mask = torch.from_numpy(g["amb/label"])[:, None] == torch.from_numpy(g["amb/label"])[torch.from_numpy(g["amb/nidx"]).long()]
    assert torch.equal(posmask.cpu(), want_mask)
    a = ops.ambiguity(p, posmask, nidx, "Method2", 0.04)
    # same fp32 operation order as the reference's CPU evaluation: agreement to a few ulp of pow/exp
    np.testing.assert_allclose(a.cpu().numpy(), g["amb/a"], rtol=0, atol=2e-6)
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.AMContrast3D.AEF.ambiguity import ambiguity_function
    a2, shares = ambiguity_function(p, posmask, 23, nidx, "Method2", 0.04, False, 0.5)
    assert torch.equal(a, a2)
    np.testing.assert_allclose(np.array(list(shares)), g["amb/shares"], rtol=0, atol=0.05)


@pytest.mark.parametrize("mode", ["Method1", "Method2", "Method3"])
def test_ambiguity_modes_vs_oracle_formula(mode):
    from amcontrast3d_amd import ops, synthetic
    sc = synthetic.make_scene(21, 5000)
    p = torch.from_numpy(sc["pos"])
    lab = torch.from_numpy(sc["y"])
    from oracle import pointops_ref as K
    o = torch.tensor([5000], dtype=torch.int32)
    nidx = K.knnquery(24, p, p, o, o)[0][:, 1:].contiguous()
    posmask = lab[:, None] == lab[nidx.long()]
    # torch-CPU composition of AEF/ambiguity.py:11-71 for all three cctype variants
    mask_num = posmask.int().sum(-1)
    top = mask_num.max()
    want = torch.abs(mask_num - top).div(top)
    b = (0 < mask_num) & (mask_num < top)
    mb = posmask[b]
    src, dst = p[b].unsqueeze(1), p[nidx[b].long()]
    dd = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dd += torch.sum(src ** 2, -1).view(-1, 1, 1)
    dd += torch.sum(dst ** 2, -1).view(dst.shape[0], 1, -1)
    dd = dd.squeeze(1)
    if mode == "Method3":
        dd = torch.sqrt(torch.abs(dd) + 1e-12)
    dpos, dneg = (mb.int() * dd).sum(-1), ((1 - mb.int()) * dd).sum(-1)
    if mode == "Method1":
        dpos, dneg = torch.full_like(dpos, 5.0), torch.full_like(dneg, 5.0)
    cc = mb.int().sum(-1) / dpos - (1 - mb.int()).sum(-1) / dneg
    want[b] = 1 / (1 + torch.full(cc.shape, np.e).pow(0.04 * cc))
    got = ops.ambiguity(p.to(DEV), posmask.to(DEV), nidx.to(DEV), mode, 0.04)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=3e-6)


@pytest.mark.parametrize("kr,ncls", [(4, 13), (16, 13), (64, 21), (100, 5)])
def test_vote_labels(kr, ncls):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(kr)
    n0, m = 5000, 777
    labels0 = torch.randint(0, ncls, (n0,), generator=g)
    nbr = torch.randint(0, n0, (m, kr), generator=g, dtype=torch.int32)
    want = torch.argmax(F.one_hot(labels0, ncls)[nbr.long()].float().mean(-2), -1)  # AEF/utils.py:39-41 + arg-max
    got = ops.vote_labels(labels0.to(torch.int32).to(DEV), nbr.to(DEV), ncls)
    assert torch.equal(got.cpu().long(), want)


@pytest.mark.parametrize("m", [1, 63, 256, 257, 5000])
@pytest.mark.parametrize("frac", [0.0, 0.3, 1.0])
def test_select_anchors(m, frac):
    """the compact list of the anchors with 0 < a <= 1 (MarginContrast.py:250-252) == torch.nonzero of the mask"""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(m)
    a = torch.rand(m, generator=g) * 1.2  # some above 1: excluded
    a[torch.rand(m, generator=g) >= frac] = 0.0
    want = torch.nonzero((0 < a) & (a <= 1)).flatten()
    sel = ops.select_anchors(a.to(DEV)).cpu()
    assert int(sel[0]) == want.numel()
    assert torch.equal(sel[1:1 + want.numel()].long(), want)


@pytest.mark.parametrize("listed", [True, False, "rev", "mutual"])
@pytest.mark.parametrize("m,C,seed", [(3000, 32, 0), (1500, 64, 1), (800, 128, 2), (300, 256, 3), (200, 20, 4),
                                      (400, 16, 5), (100, 512, 6),
                                      # >= 16384 anchors: one wave per anchor forward, one lane group per anchor backward
                                      # (below: the whole workgroup / several groups share an anchor)
                                      (40000, 32, 7), (17000, 64, 8)])
def test_contrast_stage_forward_backward(m, C, seed, listed):
    _contrast_case(m, C, seed, listed, 24)


@pytest.mark.parametrize("m,C,K", [(500, 32, 41), (300, 64, 70), (300, 128, 5), (200, 24, 37)])
def test_contrast_stage_other_neighbourhood_sizes(m, C, K):
    """k beyond one group of lanes (the backward walks the neighbours in chunks), and very small k"""
    _contrast_case(m, C, K, True, K)
    _contrast_case(m, C, K + 1, "rev", K)
    if C in (32, 64, 128) and K <= 65:
        _contrast_case(m, C, K + 2, "mutual", K)


def _contrast_case(m, C, seed, listed, K):
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(m, C, generator=g)
    idx24 = torch.randint(0, m, (m, K), generator=g, dtype=torch.int32)
    lab = torch.randint(0, 4, (m,), generator=g)
    a = torch.rand(m, generator=g)
    a[torch.rand(m, generator=g) < 0.5] = 0.0  # consistent points: excluded
    a[torch.rand(m, generator=g) < 0.05] = 1.0
    nidx = idx24[:, 1:]
    posmask = lab[:, None] == lab[nidx.long()]
    mu, nu, T = -1.0, 0.5, 0.3

    # reference composition (MarginContrast.py:250-257, 117-174) in torch on the CPU
    fr = f.clone().requires_grad_(True)
    keep = (0 < a) & (a <= 1)
    nf = fr[nidx.reshape(-1).long()].view(m, K - 1, C)
    sim = F.cosine_similarity(fr[keep].unsqueeze(-2), nf[keep], dim=2)
    pm = posmask[keep]
    margin = mu * a[keep].unsqueeze(-1) + nu
    s = (sim - margin) * pm + sim * ~pm
    e = torch.exp(s / T)
    want = (-torch.log((e * pm).sum(-1) / e.sum(-1) + 1e-12)).mean()
    (want * 0.9).backward()

    fg = f.to(DEV).requires_grad_(True)
    idx_dev = idx24.to(DEV)
    anchors = ops.select_anchors(a.to(DEV)) if listed else None
    rev = ops.contrast_csr(idx_dev[:, 1:], anchors) if listed == "rev" else None
    if rev is not None:  # the lists against a plain enumeration of the selected anchors' edges
        start, edge = rev[:m + 1].cpu().long(), rev[m + 1:].cpu().long()
        sel_rows = torch.nonzero(keep).flatten()
        pos = (sel_rows[:, None] * (K - 1) + torch.arange(K - 1)[None, :]).flatten()
        tgt = nidx[sel_rows].long().flatten()
        order = torch.argsort(tgt * (m * K) + pos)
        assert torch.equal(start, torch.searchsorted(tgt[order], torch.arange(m + 1)))
        assert torch.equal(edge[:pos.numel()], pos[order])
    mutual = None
    if listed == "mutual":  # the default of the model's plan: a multiplicity per edge + reverse lists of the non-mutual edges
        mutual, rev = ops.contrast_mutual(idx_dev[:, 1:], a.to(DEV))
        mu_cpu, start, edge = (mutual & 0x7f).cpu().long(), rev[:m + 1].cpu().long(), rev[m + 1:].cpu().long()
        want_mut = torch.zeros(m, K - 1, dtype=torch.long)
        nonmut = []
        for i in range(m if m <= 1500 else 0):  # the structure against a plain enumeration (small cases)
            row = nidx[i].long()
            for sl in range(K - 1):
                x = int(row[sl])
                cnt = int((nidx[x].long() == i).sum())
                if not bool((row[:sl] == x).any()):
                    want_mut[i, sl] = cnt
                if cnt == 0 and bool(keep[i]):
                    nonmut.append((x, i * (K - 1) + sl))
        if m <= 1500:
            assert torch.equal(mu_cpu, want_mut)
            nonmut.sort()
            assert int(start[-1]) == len(nonmut) and edge[:len(nonmut)].tolist() == [p for _, p in nonmut]
            assert torch.equal(start, torch.searchsorted(torch.tensor([x for x, _ in nonmut], dtype=torch.long), torch.arange(m + 1)))
    got = ops.contrast_stage(fg, idx_dev[:, 1:], posmask.to(DEV).contiguous(), a.to(DEV), mu, nu, T, anchors, rev, mutual)
    assert abs(float(got) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    (got * 0.9).backward()
    err = float((fg.grad.cpu() - fr.grad).norm() / fr.grad.norm())
    assert err <= 2e-5, err


def test_contrast_backward_over_mutual_edges_on_a_real_knn_graph():
    """The model's default since round 3: on the 24-NN graph of a synthetic batch (the flattened batch is one segment, as
    the loss searches it) ~90 % of the edges are mutual; the gather over mutual edges + the reverse lists of the rest gives the
    gradient of the float-atomic form, bit-reproducibly."""
    from amcontrast3d_amd import ops, synthetic
    nb = synthetic.make_batch(2, 6000, first_id=33)
    p = torch.from_numpy(nb["pos"]).reshape(-1, 3).contiguous().to(DEV)
    y = torch.from_numpy(nb["y"]).reshape(-1).to(DEV)
    m = p.shape[0]
    o = torch.tensor([m], dtype=torch.int32, device=DEV)
    idx, d2 = ops.knnquery(24, p, p, o, o)
    nidx = idx[:, 1:]
    posmask = ops.posmask_from_labels(y.int(), nidx)
    g = torch.Generator().manual_seed(3)
    a = torch.rand(m, generator=g).to(DEV)
    a[torch.rand(m, generator=g).to(DEV) < 0.3] = 0.0
    for C in (32, 64, 128, 256):
        f = torch.randn(m, C, generator=g).to(DEV)
        anchors = ops.select_anchors(a)
        mutual, rev = ops.contrast_mutual(nidx, a)
        mutual_d, rev_d = ops.contrast_mutual(nidx, a, d2[:, 1:])  # membership by distance comparison: the same structure
        assert torch.equal(mutual, mutual_d) and torch.equal(rev[:m + 1 + int(rev[m])], rev_d[:m + 1 + int(rev_d[m])])
        share = float(((mutual & 0x7f) > 0).float().mean())
        assert 0.8 < share < 1.0 and int((mutual & 0x7f).max()) == 1, share
        grads = []
        for kind in ("atomic", "mutual", "mutual"):
            fg = f.clone().requires_grad_(True)
            loss = ops.contrast_stage(fg, nidx, posmask, a, -1.0, 0.5, 0.3, anchors, rev if kind == "mutual" else None,
                                      mutual if kind == "mutual" else None)
            (loss * 0.9).backward()
            grads.append(fg.grad.clone())
        assert torch.equal(grads[1], grads[2]), "the gather form is bit-reproducible"
        err = float((grads[1] - grads[0]).norm() / grads[0].norm())
        assert err <= 2e-6, (C, err)
    print(f"mutual share of the 24-NN edges: {share:.3f}; non-mutual edges listed: {int(rev[m])}")


@pytest.mark.parametrize("B,n", [(2, 6000), (3, 1000), (1, 77)])
def test_contrast_stage_on_channel_major_embeddings(B, n):
    """ops.contrast_stage_cm reads the decoder's (B, C, n) tensor as it is; the loss is bit for bit that of contrast_stage on
    flatten(f.transpose(1, 2)) (pointnext_AA.py:518-519) with the mutual-edge plan (same unit rows, same forward kernel), the
    gradient equal to rounding and bit-reproducible.  n = 77, 1000: tiles that end inside a cloud."""
    from amcontrast3d_amd import ops, synthetic
    nb = synthetic.make_batch(B, n, first_id=5)
    p = torch.from_numpy(nb["pos"]).reshape(-1, 3).contiguous().to(DEV)
    y = torch.from_numpy(nb["y"]).reshape(-1).to(DEV)
    m = p.shape[0]
    o = torch.tensor([m], dtype=torch.int32, device=DEV)
    idx, d2 = ops.knnquery(24, p, p, o, o)
    nidx = idx[:, 1:]
    posmask = ops.posmask_from_labels(y.int(), nidx)
    g = torch.Generator().manual_seed(B * n)
    a = torch.rand(m, generator=g).to(DEV)
    a[torch.rand(m, generator=g).to(DEV) < 0.3] = 0.0
    anchors = ops.select_anchors(a)
    mutual, rev = ops.contrast_mutual(nidx, a)
    for C in (16, 32, 64, 128, 256):
        f_cm = torch.randn(B, C, n, generator=g).to(DEV)
        assert ops.contrast_stage_supported_cm(f_cm, anchors, rev, mutual)
        fa = f_cm.clone().requires_grad_(True)
        la = ops.contrast_stage_cm(fa, nidx, posmask, a, -1.0, 0.5, 0.3, anchors, rev, mutual)
        (la * 0.7).backward()
        fb = f_cm.clone().requires_grad_(True)
        lb = ops.contrast_stage(torch.flatten(fb.transpose(1, 2), 0, 1).contiguous(), nidx, posmask, a, -1.0, 0.5, 0.3, anchors, rev, mutual)
        (lb * 0.7).backward()
        assert torch.equal(la, lb), (C, float(la), float(lb))
        # (the channel-major stage builds the backward's per-anchor records from the two sums the forward pass kept, the row
        #  stage re-sums the stored cosines in another order: the gradients agree to rounding, not to the bit)
        err = float((fa.grad - fb.grad).abs().max()) / float(fb.grad.abs().max())
        assert err <= 2e-6, (C, err)
        fa2 = f_cm.clone().requires_grad_(True)
        (ops.contrast_stage_cm(fa2, nidx, posmask, a, -1.0, 0.5, 0.3, anchors, rev, mutual) * 0.7).backward()
        assert torch.equal(fa.grad, fa2.grad), "bit-reproducible"
    assert not ops.contrast_stage_supported_cm(torch.randn(B, 20, n, device=DEV), anchors, rev, mutual)  # width without row kernels


def test_contrast_stage_no_positive_anchor_is_constant():
    """n+ = 0 -> a = 1: the anchor contributes -log(1e-12) and no gradient."""
    from amcontrast3d_amd import ops
    m, C = 64, 32
    f = torch.randn(m, C, device=DEV, requires_grad=True)
    idx = torch.randint(0, m, (m, 24), dtype=torch.int32, device=DEV)
    posmask = torch.zeros(m, 23, dtype=torch.bool, device=DEV)
    a = torch.ones(m, device=DEV)
    loss = ops.contrast_stage(f, idx[:, 1:], posmask, a, -1.0, 0.5, 0.3)
    assert abs(float(loss) - 27.631021) < 1e-4
    loss.backward()
    assert float(f.grad.abs().max()) == 0.0


@pytest.mark.parametrize("B,C,N,ignored", [(2, 13, 1000, 0.0), (3, 20, 777, 0.3), (1, 2, 5, 0.0), (2, 13, 64, 1.0)])
def test_cross_entropy_matches_torch(B, C, N, ignored):
    """fused CE on (B,C,N) logits == nn.CrossEntropyLoss() on the transposed (B*N,C) copy (loss/build.py:338-340),
    value and gradient, including ignore_index = -100 targets (the ScanNet-shaped set) and the all-ignored NaN."""
    from amcontrast3d_amd import ops
    g = torch.Generator().manual_seed(C * 100 + N)
    logits = (torch.randn(B, C, N, generator=g) * 3).to(DEV)
    target = torch.randint(0, C, (B, N), generator=g).to(DEV)
    if ignored > 0:
        target[torch.rand(B, N, generator=g).to(DEV) < ignored] = -100
    lg = logits.clone().requires_grad_(True)
    lr = logits.clone().requires_grad_(True)
    got = ops.cross_entropy_mean(lg, target, -100)
    want = torch.nn.CrossEntropyLoss()(lr.transpose(1, 2).reshape(-1, C), target.flatten())
    if ignored >= 1.0:
        assert torch.isnan(got) and torch.isnan(want)
        return
    assert abs(float(got) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    (got * 0.7).backward()
    (want * 0.7).backward()
    assert float((lg.grad - lr.grad).abs().max()) <= 1e-6 * max(1.0, float(lr.grad.abs().max()) * 10)


@pytest.mark.parametrize("B,C,N,ignore", [(8, 13, 24000, None), (2, 20, 5000, -100), (1, 13, 77, None), (3, 5, 1000, 255)])
def test_confusion_matrix_from_logits_is_update_of_the_argmax(B, C, N, ignore):
    """ConfusionMatrix.update_from_logits (one launch: csrc/loss.hip confusion_kernel) == update(logits.argmax(dim=1), target),
    the call the trainer makes every iteration (main_AA.py:414-415), including ignored labels, equal logits (first maximum)
    and an out-of-range label."""
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.utils import ConfusionMatrix
    g = torch.Generator().manual_seed(B * 1000 + C)
    logits = torch.randn(B, C, N, generator=g)
    logits[:, :, ::7] = logits[:, :1, ::7]          # all classes equal: the first wins
    logits[:, 2, 1::11] = logits[:, 4 % C, 1::11]   # two-way ties
    target = torch.randint(0, C, (B, N), generator=g)
    if ignore is not None:
        target[torch.rand(B, N, generator=g) < 0.2] = ignore
    logits, target = logits.to(DEV), target.to(DEV)
    a, b = ConfusionMatrix(C, ignore), ConfusionMatrix(C, ignore)
    for _ in range(2):
        a.update(logits.argmax(dim=1), target)
        b.update_from_logits(logits, target)
    assert torch.equal(a.value, b.value) and int(a.invalid) == int(b.invalid) == 0
    for x, y in zip(a.all_metrics(), b.all_metrics()):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))
    target[0, 3] = C + 5  # outside the class range: counted, not binned
    a.update(logits.argmax(dim=1), target)
    b.update_from_logits(logits, target)
    assert torch.equal(a.value, b.value) and int(a.invalid) == int(b.invalid) == 1


def test_regression_term_as_a_two_stage_mean():
    """loss/build.py _l1_mean: nn.L1Loss's value and gradient (the AMContrast3D++ regression term over ~255000 predicted
    ambiguities) without torch's multi-block reduction, whose counters are zeroed by a memset -- a memset NODE under graph
    capture; capturable, and equal replay after replay"""
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from openpoints.loss.build import _l1_mean
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    for n in (255000, 4097, 1000):
        a = torch.rand(n, generator=g).to(dev).requires_grad_(True)
        b = torch.rand(n, generator=g).to(dev)
        got = _l1_mean(torch.nn.L1Loss(), a, b)
        got.backward()
        a2 = a.detach().clone().requires_grad_(True)
        want = torch.nn.L1Loss()(a2, b)
        want.backward()
        assert abs(float(got) - float(want)) <= 2e-7 * float(want)
        torch.testing.assert_close(a.grad, a2.grad, rtol=1e-6, atol=0)
    x, y = torch.rand(255000, device=dev), torch.rand(255000, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _l1_mean(torch.nn.L1Loss(), x, y)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = _l1_mean(torch.nn.L1Loss(), x, y)
        vals = []
        for _ in range(5):
            graph.replay()
            vals.append(float(out))
    assert len(set(vals)) == 1 and abs(vals[0] - float((x - y).abs().double().mean())) <= 1e-6
