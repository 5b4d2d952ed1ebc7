"""world_size-2 gloo rehearsal of the data-parallel plumbing bench.py uses at N > 1 (CPU only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from amcontrast3d_amd import dist as adist
    r, l, w = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    ids = adist.scene_ids(r, w, 4)
    # slowest rank defines the step time
    t = adist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # gradient averaging: flat buckets == per-tensor mean over ranks
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    x = torch.randn(6, 8, generator=torch.Generator().manual_seed(100 + rank))
    net(x).square().mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    nb = adist.allreduce_gradients(list(net.parameters()), bucket_bytes=256)
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.tolist() for g in local])
    want = [sum(torch.tensor(gathered[r_][i]) for r_ in range(world)) / world for i in range(len(local))]
    ok = all(torch.allclose(p.grad, w_, atol=1e-6) for p, w_ in zip(net.parameters(), want))
    # flat gradient buffer: backward accumulates into the views in place; one all-reduce averages everything
    torch.manual_seed(0)
    net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    net2.load_state_dict(net.state_dict())
    fg = adist.FlatGradients(list(net2.parameters()))
    net2(x).square().mean().backward()   # a first pass that zero() must erase
    fg.zero()
    net2(x).square().mean().backward()
    flat_ok = fg.intact() and all(torch.allclose(p.grad, g, atol=1e-7) for p, g in zip(net2.parameters(), local))
    fg.allreduce()
    flat_ok = flat_ok and fg.intact() and all(torch.allclose(p.grad, w_, atol=1e-6) for p, w_ in zip(net2.parameters(), want))
    torch.optim.SGD(net2.parameters(), lr=0.1).zero_grad(set_to_none=True)
    flat_ok = flat_ok and not fg.intact()  # detects a replaced .grad
    # copy mode: backward with empty .grad slots, then one multi-tensor copy into the buffer
    fc = adist.FlatGradients(list(net2.parameters()), accumulate=False)
    for _ in range(2):  # the second pass must not see the first one's gradients
        fc.zero()
        net2(x).square().mean().backward()
        fc.gather()
    flat_ok = flat_ok and fc.intact() and all(torch.allclose(p.grad, g, atol=1e-7) for p, g in zip(net2.parameters(), local))
    fc.allreduce()
    flat_ok = flat_ok and fc.intact() and all(torch.allclose(p.grad, w_, atol=1e-6) for p, w_ in zip(net2.parameters(), want))
    ok = ok and flat_ok
    # DDP wrapper (CPU branch) keeps replicas in sync after a step
    ddp = adist.wrap_data_parallel(torch.nn.Linear(4, 2), torch.device("cpu"), world)
    opt = torch.optim.SGD(ddp.parameters(), lr=0.1)
    ddp(torch.randn(3, 4, generator=torch.Generator().manual_seed(rank))).sum().backward()
    opt.step()
    w0 = [None] * world
    dist.all_gather_object(w0, ddp.module.weight.detach().tolist())
    adist.barrier()
    q.put((rank, ids, t, nb, ok, w0[0] == w0[1]))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ids0, t0, nb0, ok0, same0), (r1, ids1, t1, nb1, ok1, same1) = res
    assert ids0 == [0, 1, 2, 3] and ids1 == [4, 5, 6, 7]  # disjoint scene shards, fixed per-rank batch
    assert t0 == t1 == 2.0
    assert nb0 == nb1 and nb0 > 1 and ok0 and ok1 and same0 and same1


def test_single_process_defaults():
    from amcontrast3d_amd import dist as adist
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert adist.env_world() == (0, 0, 1)
    assert adist.max_over_ranks(3.5, torch.device("cpu")) == 3.5
    assert adist.allreduce_gradients([]) == 0
    m = torch.nn.Linear(2, 2)
    assert adist.wrap_data_parallel(m, torch.device("cpu"), 1) is m


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment: the parent (no GPU call) starts two ranks as children
    and rank 0 prints ONE line with n_gpus 2 -- here on the CPU rehearsal path (gloo, stand-in model): scene shards are
    disjoint, the flat gradient all-reduce keeps the replicas bit-identical."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-cpu", "--steps", "3",
                          "--warmup", "1", "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["replicas_in_sync"] is True and line["scene_ids_rank0"] == [0, 1, 2, 3]


def test_segmented_graph_orders_graphs_and_collectives():
    """graphs.SegmentedGraph without a GPU: collective() outside a capture just runs the call (the eager path)."""
    from amcontrast3d_amd import graphs
    seen = []
    assert graphs.collective(lambda: seen.append(1) or 7) == 7 and seen == [1]
    sg = graphs.SegmentedGraph()
    assert sg.segments == 0 and sg.collectives == 0
