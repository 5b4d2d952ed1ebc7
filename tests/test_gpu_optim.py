"""amcontrast3d_amd.fused_optim.FusedAdamW (csrc/optim.hip) against the pair it replaces in the trainer's step,
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (main_AA.py:586-592): same parameters after several steps, the same
returned gradient norm, torch's state_dict layout (checkpoints resume either way)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(32, 7, 1), (32,), (32,), (64, 32, 1, 1), (64,), (5000, 3), (1,), (13, 64, 1), (1025,), (1024,), (3, 3)]
    return [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]


def _groups(ps):
    return [{"params": [p for p in ps if p.dim() == 1], "weight_decay": 0.0},
            {"params": [p for p in ps if p.dim() != 1], "weight_decay": 1e-2, "lr": 3e-3}]


@pytest.mark.parametrize("clip", [None, 0.5, 1e6])
def test_matches_clip_grad_norm_plus_torch_adamw(clip):
    from amcontrast3d_amd.fused_optim import FusedAdamW
    a, b = _params(0), _params(0)
    ref = torch.optim.AdamW(_groups(a), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    got = FusedAdamW(_groups(b), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        grads = [torch.randn(p.shape, generator=g).to(DEV) * (0.1 if it % 2 else 3.0) for p in a]
        for p, q, gr in zip(a, b, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 3:  # a parameter without a gradient is skipped by both
            a[2].grad = b[2].grad = None
        want_norm = torch.nn.utils.clip_grad_norm_(a, clip, norm_type=2) if clip else None
        ref.step()
        norm = got.step(max_grad_norm=clip)
        if clip:
            assert abs(float(norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        for p, q in zip(a, b):
            assert float((p - q).detach().abs().max()) <= 2e-6 * max(1.0, float(p.detach().abs().max())), (it, tuple(p.shape))
    sa, sb = ref.state_dict(), got.state_dict()
    assert sa["param_groups"][1]["lr"] == sb["param_groups"][1]["lr"] == 3e-3
    for k in sa["state"]:
        for name in ("exp_avg", "exp_avg_sq"):
            x, y = sa["state"][k][name], sb["state"][k][name]
            assert float((x - y).abs().max()) <= 1e-6 * max(1e-3, float(x.abs().max())), (k, name)
        assert float(sb["state"][k]["step"]) == float(sa["state"][k]["step"])  # per parameter: one of them skipped a step


def test_large_model_two_level_norm():
    """more than 2048 chunks of 1024 elements (PointNeXt-XL has 40 k): the chunk norms are folded by an extra launch"""
    from amcontrast3d_amd.fused_optim import FusedAdamW
    g = torch.Generator().manual_seed(2)
    shapes = [(1024, 1024), (777, 513), (2048, 700, 1), (4097,), (1,)]
    a = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref = torch.optim.AdamW(a, lr=1e-2, weight_decay=1e-2)
    got = FusedAdamW(b, lr=1e-2, weight_decay=1e-2)
    for it in range(3):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad, q.grad = gr.clone(), gr.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(a, 10.0, norm_type=2)
        ref.step()
        norm = got.step(max_grad_norm=10.0)
        assert abs(float(norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        for p, q in zip(a, b):
            assert float((p - q).detach().abs().max()) <= 2e-6 * max(1.0, float(p.detach().abs().max()))


def test_state_dict_round_trip_and_resume_from_torch_adamw():
    from amcontrast3d_amd.fused_optim import FusedAdamW
    a, b, c = _params(3), _params(3), _params(3)
    ref = torch.optim.AdamW(_groups(a), lr=1e-2)
    g = torch.Generator().manual_seed(4)
    grads = [[torch.randn(p.shape, generator=g).to(DEV) for p in a] for _ in range(4)]
    for gs in grads[:2]:
        for p, gr in zip(a, gs):
            p.grad = gr.clone()
        ref.step()
    # resume the fused optimizer from torch's checkpoint, continue both
    for p, q in zip(a, b):
        q.data.copy_(p.data)
    got = FusedAdamW(_groups(b), lr=1e-2)
    got.load_state_dict(copy.deepcopy(ref.state_dict()))
    for gs in grads[2:]:
        for p, q, gr in zip(a, b, gs):
            p.grad, q.grad = gr.clone(), gr.clone()
        ref.step()
        got.step()
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max()))
    # ... and torch's AdamW from the fused optimizer's checkpoint
    for p, q in zip(b, c):
        q.data.copy_(p.data)
    back = torch.optim.AdamW(_groups(c), lr=1e-2)
    back.load_state_dict(copy.deepcopy(got.state_dict()))
    for p, q in zip(b, c):
        p.grad, q.grad = grads[0][0].new_ones(p.shape), grads[0][0].new_ones(p.shape)
    got.step()
    back.step()
    for p, q in zip(b, c):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max()))


def test_replays_in_a_hip_graph():
    """the step as the trainer's captured update: gradients arrive in static tensors, the table is built by one eager step"""
    from amcontrast3d_amd.fused_optim import FusedAdamW
    a, b = _params(5), _params(5)
    ref, got = torch.optim.AdamW(_groups(a), lr=1e-2), FusedAdamW(_groups(b), lr=1e-2)
    static = [torch.zeros_like(p) for p in b]
    for q, s in zip(b, static):
        q.grad = s
    for p in a:
        p.grad = torch.zeros_like(p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        got.step(max_grad_norm=1.0)           # eager, zero gradients: moments stay zero, parameters only decay
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            got.step(max_grad_norm=1.0)
    torch.cuda.synchronize()
    torch.nn.utils.clip_grad_norm_(a, 1.0)
    ref.step()
    g = torch.Generator().manual_seed(6)
    for it in range(3):
        grads = [torch.randn(p.shape, generator=g).to(DEV) for p in a]
        for p, s, gr in zip(a, static, grads):
            p.grad = gr.clone()
            s.copy_(gr)
        torch.nn.utils.clip_grad_norm_(a, 1.0)
        ref.step()
        torch.cuda.synchronize()
        graph.replay()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 3e-6 * max(1.0, float(p.abs().max()))


def test_resume_from_a_torch_checkpoint_with_a_stateless_parameter_and_zero_grad_keeps_addresses():
    """torch.optim.AdamW creates state lazily: a parameter that never had a gradient has no entry in its checkpoints
    (ADVICE r2).  The fused optimizer restarts that parameter's moments from zero and keeps stepping; zero_grad() zeroes in
    place, so the device table (which holds every .grad address) is not rebuilt from step to step."""
    from amcontrast3d_amd.fused_optim import FusedAdamW
    a, b = _params(9), _params(9)
    ref = torch.optim.AdamW(_groups(a), lr=1e-2)
    g = torch.Generator().manual_seed(10)
    for _ in range(2):
        for i, p in enumerate(a):
            p.grad = None if i == 4 else torch.randn(p.shape, generator=g).to(DEV)
        ref.step()
    sd = copy.deepcopy(ref.state_dict())
    assert len(sd["state"]) == len(a) - 1
    for p, q in zip(a, b):
        q.data.copy_(p.data)
    got = FusedAdamW(_groups(b), lr=1e-2)
    got.load_state_dict(sd)
    tables = []
    for _ in range(3):
        grads = [torch.randn(p.shape, generator=g).to(DEV) for p in a]
        for p, q, gr in zip(a, b, grads):
            p.grad = gr.clone()
            if q.grad is None:
                q.grad = gr.clone()
            else:
                q.grad.copy_(gr)
        ref.step()
        got.step()
        tables.append(got._table.data_ptr())
        got.zero_grad()
        assert all(q.grad is not None and float(q.grad.abs().max()) == 0.0 for q in b)
    assert len(set(tables)) == 1, "the device table was rebuilt although no gradient tensor moved"
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max()))
