"""Cross-rank BatchNorm on the fused kernels (ops.SyncBatchNormFused) = torch.nn.SyncBatchNorm, which the reference
turns every BN layer into when world_size > 1 (examples/segmentation/main_AA.py:146-148, 820).

Two ranks share the one GPU of the test box over gloo (RCCL refuses two ranks on one device); each holds half of a
batch, and must reproduce (a) the single-process fused BatchNorm on the whole batch and (b) torch's own
SyncBatchNorm -> ReLU [-> max] on its half.  A one-rank RCCL group covers the all-reduce under hipGraph capture."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

TOL = 2e-5  # fp32 normalisation from fp64 sums; the sums themselves are added in another order across ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(pool, seed=0, odd=False):
    g = torch.Generator().manual_seed(seed)
    if odd:  # lengths that are not multiples of 4 (scalar load paths), K = 20 neighbours
        shape = (4, 24, 37, 20) if pool else (4, 24, 333)
    else:
        shape = (4, 24, 50, 32) if pool else (4, 24, 1000)
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma = torch.rand(24, generator=g) + 0.5
    beta = torch.randn(24, generator=g) * 0.2
    gshape = shape[:-1] if pool else shape
    gout = torch.randn(gshape, generator=g)
    return x, gamma, beta, gout


def _run_fused(fn, x, gamma, beta, gout, *extra):
    x = x.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    y = fn(x, gamma, beta, 1e-5, True, *extra)[0]
    y.backward(gout)
    return y.detach(), x.grad, gamma.grad, beta.grad


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import torch.distributed as dist
    from amcontrast3d_amd import ops
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    out = {}
    for pool, odd in ((False, False), (True, False), (False, True), (True, True)):
        x, gamma, beta, gout = [t.to(dev) for t in _case(pool, odd=odd)]
        # ranks of unequal size in the odd cases: 3 + 1 clouds (element counts are exchanged, not assumed)
        half = (slice(0, 3) if rank == 0 else slice(3, 4)) if odd else slice(2 * rank, 2 * rank + 2)
        bn = torch.nn.BatchNorm2d(24) if pool else torch.nn.BatchNorm1d(24)
        bn = bn.to(dev)
        y, dx, dg, db = _run_fused(ops.SyncBatchNormFused.apply, x[half].contiguous(), gamma, beta, gout[half].contiguous(),
                                   pool, bn, dist.group.WORLD)
        # (a) the whole batch in one process
        plain = ops.BatchNormMax if pool else ops.BatchNormAct
        bn_full = (torch.nn.BatchNorm2d(24) if pool else torch.nn.BatchNorm1d(24)).to(dev)
        yf, dxf, dgf, dbf = _run_fused(plain.apply, x, gamma, beta, gout, bn_full)
        both = torch.stack([dg, db])
        dist.all_reduce(both)  # parameter gradients are rank-local sums
        # (b) torch's SyncBatchNorm on this rank's half
        sbn = torch.nn.SyncBatchNorm(24).to(dev)
        with torch.no_grad():
            sbn.weight.copy_(gamma)
            sbn.bias.copy_(beta)
        xt = x[half].clone().requires_grad_(True)
        yt = torch.relu(sbn(xt))
        if pool:
            yt = yt.max(dim=-1)[0]
        yt.backward(gout[half])
        out[(pool, odd)] = dict(
            y_full=float((y - yf[half]).abs().max()), dx_full=float((dx - dxf[half]).abs().max()),
            dg_full=float((both[0] - dgf).abs().max() / dgf.abs().max()),
            db_full=float((both[1] - dbf).abs().max() / dbf.abs().max()),
            y_torch=float((y - yt).abs().max()), dx_torch=float((dx - xt.grad).abs().max()),
            dg_torch=float((dg - sbn.weight.grad).abs().max() / dg.abs().max()),
            db_torch=float((db - sbn.bias.grad).abs().max() / db.abs().max()),
            rm=float((bn.running_mean - bn_full.running_mean).abs().max()),
            rv=float((bn.running_var - bn_full.running_var).abs().max()),
            rv_torch=float((bn.running_var - sbn.running_var).abs().max()),
            tracked=int(bn.num_batches_tracked))
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_ranks_match_whole_batch_and_torch_syncbn():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        for key, r in res[rank].items():
            assert r.pop("tracked") == 1
            for k, v in r.items():
                assert v <= TOL, (rank, key, k, v)


def _model_worker(rank, world, port, q, variant="S", kw=None):
    """PointNeXt-S with every BN converted to SyncBatchNorm (as main_AA.py:146-148 does), one cloud per rank, against
    the plain model on both clouds in one process: same logits; parameter gradients of a sum-type objective add up."""
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import torch.distributed as dist
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import configs, synthetic
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    c = EasyConfig()
    c.update(configs.model_cfg(variant, dropout=0, **(kw or {})))
    torch.manual_seed(0)
    plain = build_model_from_cfg(c).to(dev).train()
    c2 = EasyConfig()
    c2.update(configs.model_cfg(variant, dropout=0, **(kw or {})))
    synced = build_model_from_cfg(c2).to(dev).train()
    synced.load_state_dict(plain.state_dict())
    synced = torch.nn.SyncBatchNorm.convert_sync_batchnorm(synced)
    nb = synthetic.make_batch(2, 2048, first_id=40)
    full = {k: torch.from_numpy(v).to(dev) for k, v in nb.items()}
    mine = {k: v[rank:rank + 1].contiguous() for k, v in full.items()}
    probe = torch.randn(2, 13, 2048, generator=torch.Generator().manual_seed(5)).to(dev)
    lf = plain(full)[0]
    (lf * probe).sum().backward()
    ls = synced(mine)[0]
    (ls * probe[rank:rank + 1]).sum().backward()
    n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in synced.modules())
    worst_g, worst_name = 0.0, None
    gmax = max(float(pp.grad.norm()) for pp in plain.parameters())
    for (name, pp), ps in zip(plain.named_parameters(), synced.parameters()):
        g = ps.grad.clone()
        dist.all_reduce(g)
        # a ~zero gradient (a conv bias in front of a BatchNorm) is judged on the scale of the others
        err = float((g - pp.grad).norm() / max(float(pp.grad.norm()), 1e-3 * gmax))
        if err > worst_g:
            worst_g, worst_name = err, name
    stats = max(float((a.running_var - b.running_var).abs().max())
                for a, b in zip([m for m in plain.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)],
                                [m for m in synced.modules() if isinstance(m, torch.nn.SyncBatchNorm)]))
    scale = float(lf.abs().max())
    q.put((rank, float((ls - lf[rank:rank + 1]).abs().max()) / scale, worst_g, worst_name, stats, n_sync))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant,kw,nbn", [("S", None, 17), ("L", {"width": 16, "blocks": [1, 2, 2, 1, 1]}, 19)])
def test_model_with_synced_bn_equals_whole_batch_model(variant, kw, nbn):
    """PointNeXt-S (GroupedConvBN + the recomputing tail / bn_max under SyncBatchNorm) and a model with InvResMLP blocks
    (LocalAggregationFused): the convolve-before-gather layers exchange their statistics between two phases"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q, variant, kw)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, logit_err, grad_err, grad_name, stat_err, n_sync in res:
        assert n_sync == nbn, n_sync
        assert logit_err <= 1e-4, (rank, logit_err)
        assert stat_err <= 1e-4, (rank, stat_err)
        # arg-max routing of the neighbourhood max-pool flips on near-ties (tests/test_gpu_model.py: GRAD_RTOL)
        assert grad_err <= 3e-2, (rank, grad_name, grad_err)


def _graph_worker(port, q, segmented=False):
    os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import torch.distributed as dist
    from amcontrast3d_amd import ops
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    x, gamma, beta, gout = [t.to(dev) for t in _case(True, seed=3)]
    bn = torch.nn.BatchNorm2d(24).to(dev)
    want = _run_fused(ops.BatchNormMax.apply, x, gamma, beta, gout, torch.nn.BatchNorm2d(24).to(dev))
    xs = x.clone().requires_grad_(True)
    gs, bs = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)

    def step():
        xs.grad = gs.grad = bs.grad = None
        y = ops.SyncBatchNormFused.apply(xs, gs, bs, 1e-5, True, True, bn, dist.group.WORLD)[0]
        y.backward(gout)
        return y

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()  # warm-up outside capture (communicator set-up, allocator)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if segmented:
        # bench.py's N > 1 form: the two all-reduces are NOT captured; the step becomes graph | all-reduce | graph |
        # all-reduce | graph, replayed in that order (amcontrast3d_amd/graphs.py); backward stays on this thread
        from amcontrast3d_amd.graphs import SegmentedGraph
        out = {}
        with torch.cuda.stream(side):
            graph = SegmentedGraph("thread_local").capture(lambda: out.update(y=step()))
            y = out["y"]
            assert (graph.segments, graph.collectives) == (3, 2), (graph.segments, graph.collectives)
            with torch.no_grad():
                xs.copy_(x)
            graph.replay()
    else:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):  # the RCCL watchdog thread polls events meanwhile
            y = step()
        with torch.no_grad():
            xs.copy_(x)
        graph.replay()
    torch.cuda.synchronize()
    got = (y.detach(), xs.grad, gs.grad, bs.grad)
    q.put([float((a - b).abs().max()) for a, b in zip(got, want)] + [int(bn.num_batches_tracked)])
    dist.destroy_process_group()


@pytest.mark.parametrize("segmented", [False, True])
def test_one_rank_rccl_under_graph_capture(segmented):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_graph_worker, args=(_free_port(), q, segmented))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert res[-1] == 3  # two warm-ups and one replay (the capture pass itself executes nothing)
    assert all(v <= TOL for v in res[:-1]), res
