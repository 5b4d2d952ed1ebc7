#!/usr/bin/env python
"""Train-step throughput of the AMContrast3D hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): train-step points/sec -- forward + CrossEntropyAce (cross-entropy +
adaptive-margin contrast over 4 decoder stages) + backward (+ gradient all-reduce for N > 1)
+ clip + AdamW step -- on synthetic S3DIS-shaped 24 000-point clouds.  Workload at every N:
BASELINE config 2, PointNeXt-S + AMContrast3D-AA, batch 8 clouds per GPU (weak scaling: scenes are
sharded across ranks, one process per GPU, DDP + SyncBN over RCCL exactly as
examples/segmentation/main_AA.py:146-152 does).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

`--gpus N` without a torchrun environment starts the N ranks itself (a torch.distributed.run child process,
launched before this process has made any GPU call) and prints their one line.

Besides the contract fields the line carries
  roofline       the dominant native kernel ON THE STEP'S CRITICAL PATH (the feature half on the main stream;
                 largest share of HIP-event time among its C-ABI launches, measured live): algorithmic bytes
                 (or FLOPs) per launch / average launch duration vs 8 TB/s HBM (157.3 TFLOP/s fp32 MFMA)
  roofline_step  the whole step against both roofs: SURVEY 8(d)'s algorithmic bytes and dense FLOPs per step /
                 the measured step time, and the HBM bytes the PMC passes under profiles/ measured
  latency_chain  the FPS chain (runs two steps ahead on a queue of its own): microseconds per dependent iteration
  ms_per_step_no_overlap   the same parts replayed back to back on one stream (a joint FPS launch once per J steps)
  single_batch_latency_ms  sum of the parts alone: what ONE batch takes from raw points to updated weights
  cpu_baseline   the oracle's CPU restatement of the SAME step (oracle/model_ref.py on oracle/pointops_ref.c,
                 OpenMP + torch CPU threads) on this box's host cores, rank 0 and N = 1 only: 1 warm-up + 3 timed
                 steps (median, forward / loss / backward split) at 2 clouds, 2 timed steps at the full batch
  kernels        per-operator HIP-event totals for the timed region (ms per step)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# SURVEY.md section 8(d) / BASELINE.md section 2: algorithmic work per POINT of a train step (forward x 3), derived there
# for B=8 x N=24000 (S: 1.20 GB, 137.7 GF; L: 1.99 GB, 578.8 GF; XL: 4.26 GB, 3241 GF per 192000 points)
ALGORITHMIC_PER_POINT = {"S": (1.197e9 / 192000, 137.7e9 / 192000), "L": (1.992e9 / 192000, 578.8e9 / 192000),
                         "XL": (4.26e9 / 192000, 3241e9 / 192000)}
GEOMETRY_OPS = ("furthest_point_sampling", "ball_query", "three_nn", "knnquery", "posmask", "ambiguity", "vote_labels")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=24000)
    ap.add_argument("--variant", default="S")
    ap.add_argument("--mm", action="store_true",
                    help="AMContrast3D++ (BaseSeg_M_AMContrast3D + CrossEntropyAcePre) instead of AMContrast3D")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--fps-lanes", type=int, default=0,
                    help="future batches whose FPS runs as one joint launch (one launch every J steps on the sampling queue).  "
                         "0 = choose: 2 where the first-level chain is about a feature half long (24k-point clouds), else 3-8 "
                         "so that the chain of all levels fits into J steps (64k / 120k-point clouds in small batches: "
                         "16000-30000 dependent iterations on ONE workgroup per cloud)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="compute each batch's geometry inline instead of one step ahead on a side stream")
    ap.add_argument("--sync-bn", action="store_true", help="(default at N > 1; kept for older command lines)")
    ap.add_argument("--no-sync-bn", action="store_true",
                    help="N > 1: per-rank BatchNorm statistics.  Default at N > 1 is the reference's behaviour "
                         "(main_AA.py:146-148, 820: every BN layer becomes SyncBatchNorm): statistics over all ranks on "
                         "the fused kernels, one small all-reduce per layer and direction issued eagerly BETWEEN the "
                         "captured segments of the step (amcontrast3d_amd/graphs.py)")
    ap.add_argument("--pool", type=int, default=4,
                    help="distinct resident batches rotated through the pipeline (different geometry every step)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: the reference's use_amp (main_AA.py:389-394): model and criterion under autocast; here the 1x1 "
                         "convolutions then run on the bf16 MFMA with fp32 accumulation, tensors stay fp32")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs: stop after the timed loop (no parts-alone / serial / per-operator passes, no CPU "
                         "baseline), so that the tail of a rocprofv3 trace is the steady state")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: run only the multi-rank control flow (launcher, process group, scene shards, flat "
                         "gradient all-reduce, barrier/max timing, replica check) on a small torch CPU model over gloo; "
                         "prints a line with metric 'rehearsal' (tests/test_dist_cpu.py)")
    ap.add_argument("--ddp", action="store_true",
                    help="N > 1: torch's SyncBatchNorm + DistributedDataParallel wrappers (eager), the literal "
                         "main_AA.py:146-152 recipe, as a cross-check of the two paths above")
    ap.add_argument("--eval", action="store_true",
                    help="instead of the train step: whole-room testing (amcontrast3d_amd.evaluate, the reference's "
                         "test_boundary_inner) of a synthetic --room-points room; its own JSON line")
    ap.add_argument("--room-points", type=int, default=300000)
    ap.add_argument("--cpu-baseline-batch", type=int, default=2,
                    help="clouds in the CPU sample (bounded: the full batch of 8 takes minutes on the host)")
    return ap.parse_args()


def build(variant, dev, world, ddp, mm=False, sync_bn=False):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import configs, dist as adist
    from openpoints.loss import build_criterion_from_cfg
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig
    torch.manual_seed(0)
    cfg = configs.model_cfg_mm(variant, dropout=0.5) if mm else configs.model_cfg(variant, dropout=0.5)
    c = EasyConfig(); c.update(cfg)
    model = build_model_from_cfg(c).to(dev).train()
    if ddp:
        model = adist.wrap_data_parallel(model, dev, world)
    elif sync_bn:  # main_AA.py:146-148; blocks.run_convblocks routes these layers to ops.SyncBatchNormFused
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    cc = EasyConfig(); cc.update(configs.criterion_cfg_mm() if mm else configs.criterion_cfg())
    criterion = build_criterion_from_cfg(cc).to(dev)
    aargs = EasyConfig(); aargs.update(configs.ambiguity_args_mm("s3dis") if mm else configs.ambiguity_args("s3dis"))
    # cfgs/s3dis/default.yaml:64-72: AdamW lr 0.01 wd 1e-4 (1-d parameters and biases undecayed), clip 10
    from openpoints.optim import build_optimizer_from_cfg
    opt = build_optimizer_from_cfg(model, NAME="adamw", lr=0.01, weight_decay=1e-4)  # fused + capturable on the GPU
    return cfg, model, criterion, aargs, opt


def cpu_baseline(cfg, model, batch_np, aargs_dict, points, small=2):
    """The oracle's CPU restatement of the step on this box's host cores: 1 warm-up + 3 timed
    steps at `small` clouds (median; forward / loss / backward split), then 2 timed steps of the full batch, whose median
    is `value` (the same workload as the GPU line; ~1 minute of CPU work in all)."""
    import statistics
    from oracle import model_ref, pointops_ref
    pointops_ref.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("AMC3D_CPU_THREADS", "16")))  # the box's CPU share for one GPU
    torch.set_num_threads(cores)
    pointops_ref.set_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in (model.module if hasattr(model, "module") else model).state_dict().items()}
    cfg = json.loads(json.dumps(cfg))
    cfg["cls_args"]["dropout"] = 0

    def run(nclouds, reps, warm):
        data = {k: torch.from_numpy(v[:nclouds]) for k, v in batch_np.items()}
        rows = []
        for i in range(warm + reps):
            tm = {}
            t0 = time.perf_counter()
            model_ref.train_step(sd, cfg, data, data["y"], 13, None, aargs_dict, timings=tm)
            tm["step"] = time.perf_counter() - t0
            if i >= warm:
                rows.append(tm)
        return {k: statistics.median(r[k] for r in rows) for k in ("step", "forward", "loss", "backward")}

    full = batch_np["pos"].shape[0]
    small = min(small, full)
    ms = run(small, 3, 1)
    out = {"value": small * points / ms["step"], "unit": "points/s", "cores": cores, "kind": "port",
           "sample": f"median of 3 steps after 1 warm-up, batch {small} x {points} points: {ms['step']:.2f} s/step "
                     f"(forward {ms['forward']:.2f}, loss {ms['loss']:.2f}, backward {ms['backward']:.2f}); "
                     f"oracle/model_ref.py + pointops_ref.c, OpenMP/torch {cores} threads; no optimizer step"}
    if full > small and not os.environ.get("AMC3D_CPU_BASELINE_SMALL_ONLY"):
        mf = run(full, 2, 0)
        out["value_small_batch"] = out["value"]
        out["value"] = full * points / mf["step"]
        out["sample"] = (f"median of 2 steps, batch {full} x {points} points (the GPU line's workload): {mf['step']:.2f} s/step "
                         f"(forward {mf['forward']:.2f}, loss {mf['loss']:.2f}, backward {mf['backward']:.2f}); and "
                         + out["sample"])
    return out


def launch_ranks(args):
    """`--gpus N` without a torchrun environment: start the N ranks as a CHILD process tree (torch.distributed.run) and
    pass their output through.  This process has made no GPU call (importing torch makes none), so nothing that
    touched the GPU is ever re-executed or forked."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rehearse_cpu(args):
    """The N-rank control flow of main() on a stand-in torch CPU model over gloo (no kernels): what the CPU test suite
    can check of the multi-GPU path -- that `--gpus N` starts N ranks, shards scenes, keeps replicas in sync through
    the flat gradient all-reduce and reports max-over-ranks time on rank 0."""
    from amcontrast3d_amd import dist as adist
    rank, local, world = adist.init_from_env(backend="gloo")
    ids = adist.scene_ids(rank, world, args.batch)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    params = list(net.parameters())
    flatg = adist.FlatGradients(params, accumulate=False)
    adist.barrier()
    t0 = time.perf_counter()
    for step in range(args.warmup + args.steps):
        x = torch.randn(16, 8, generator=torch.Generator().manual_seed(1000 * ids[0] + step))
        flatg.zero()
        net(x).square().mean().backward()
        flatg.gather()
        flatg.allreduce()
        opt.step()
    adist.barrier()
    dt = adist.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    sync = True
    if world > 1:
        import torch.distributed as tdist
        chk = torch.stack([p.detach().double().sum() for p in params])
        lo, hi = chk.clone(), chk.clone()
        from amcontrast3d_amd.graphs import on_side_stream
        on_side_stream(lambda: (tdist.all_reduce(lo, op=tdist.ReduceOp.MIN), tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)))
        sync = bool(torch.equal(lo, hi))
    if rank == 0:
        print(json.dumps({"metric": "rehearsal", "value": 0.0, "unit": "none", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3),
                          "replicas_in_sync": sync, "scene_ids_rank0": ids, "data": "synthetic"}))
    if world > 1:
        torch.distributed.destroy_process_group()


def eval_main(args):
    """one room = voxel partition into sub-clouds, eval-mode model on every sub-cloud,
    mean vote per point, whole / boundary / inner confusion matrices.  A 'step' is one whole room."""
    import numpy as np
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import _lib, configs, evaluate, synthetic
    from openpoints.models import build_model_from_cfg
    from openpoints.utils import EasyConfig
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the product path)"
    dev = torch.device("cuda", 0)
    _lib.load()
    torch.manual_seed(0)
    c = EasyConfig(); c.update(configs.model_cfg(args.variant, dropout=0.5))
    model = build_model_from_cfg(c).to(dev).eval()
    room = synthetic.make_batch(1, args.room_points, first_id=900, voxel_size=0.02)
    coord = room["pos"][0] - room["pos"][0].min(0)
    feat = room["x"][0, :3].T.copy()
    label_np = room["y"][0].astype(np.int64)
    label = torch.from_numpy(label_np).to(dev)
    parts = evaluate.voxel_parts(coord, 0.04)

    def one_room():
        return evaluate.test_cloud_boundary_inner(model, coord, feat, label, parts, 13, None, 24)

    for _ in range(args.warmup):
        one_room()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_room()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    npts = len(parts) * len(parts[0])
    line = {"metric": "whole-room test sub-cloud points/sec (eval-mode model + vote + boundary/inner matrices)",
            "value": round(npts / dt, 1), "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PointNeXt-{args.variant}, one room of {args.room_points} points in {len(parts)} "
                                   f"sub-clouds of {len(parts[0])} points (voxel 0.04), inputs on the host"},
            "miou_whole_boundary_inner": [round(v, 3) for v in evaluate.summarize(r["cm"], r["cm_b"], r["cm_i"])[0:15:5]]}
    if not args.no_cpu_baseline:
        from oracle import eval_ref, pointops_ref
        pointops_ref.build()
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMC3D_CPU_THREADS", "16")))
        pointops_ref.set_threads(cores); torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        cfg = json.loads(json.dumps(configs.model_cfg(args.variant, dropout=0)))
        t0 = time.perf_counter()
        eval_ref.test_cloud(sd, cfg, coord, feat, label_np, parts[:2], 13, None, 24)
        dtc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(2 * len(parts[0]) / dtc, 1), "unit": "points/s", "cores": cores, "kind": "port",
                                "sample": f"2 of the {len(parts)} sub-clouds, {dtc:.1f} s (oracle/eval_ref.py on model_ref.py + "
                                          f"pointops_ref.c, {cores} threads)"}
    print(json.dumps(line))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # children do the work; this process never touches the GPU
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    if args.eval:
        return eval_main(args)
    from amcontrast3d_amd import _lib, configs, dist as adist, synthetic, timing
    rank, local, world = adist.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}: the environment's world size is used", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the product path)"
    local = local % torch.cuda.device_count()  # (rehearsals put several ranks on one card; a real node has one each)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()

    use_ddp = world > 1 and args.ddp
    sync_bn = world > 1 and not args.no_sync_bn and not use_ddp  # the reference's behaviour whenever distributed
    if not args.no_sync_bn and world == 1 and os.environ.get("AMC3D_FORCE_SYNC_BN"):
        # rehearsal on a one-GPU box: a one-rank RCCL group, so that the captured step contains the all-reduces
        import torch.distributed as tdist
        import amcontrast3d_amd
        amcontrast3d_amd.activate()
        from openpoints.models.layers import blocks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ["NCCL_DEBUG"] = os.environ.get("AMC3D_NCCL_DEBUG", "WARN")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        blocks._FORCE_SYNCED_BN = sync_bn = True
    use_graph = not args.no_graph and not use_ddp
    cfg, model, criterion, aargs, opt = build(args.variant, dev, world, use_ddp, args.mm, sync_bn)
    # `--pool` distinct resident batches rotate through the pipeline: every step sees another cloud geometry (k-NN tie
    # counts, grid occupancy, cache contents), as a training loop would; scene ids are disjoint across ranks and steps
    npool = max(1, args.pool)
    pool_np = [synthetic.make_batch(args.batch, args.points, first_id=adist.scene_ids(rank, world, args.batch, step=j)[0])
               for j in range(npool)]
    nb = pool_np[0]
    pool = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in pool_np]
    data = {k: v.clone() for k, v in pool[0].items()}  # the feature half's static input buffers (batch t)
    params = list(model.parameters())
    # N > 1: gradients live in one flat buffer, exchanged by a single RCCL all-reduce between the two captured halves
    flatg = (adist.FlatGradients(params, accumulate=bool(os.environ.get("AMC3D_FLAT_ACCUMULATE")))
             if (world > 1 and not use_ddp) or os.environ.get("AMC3D_FLAT_GRADS") else None)
    torch.cuda.synchronize()
    out = {}

    def fwd_bwd():
        if flatg is not None:
            flatg.zero()  # part of the captured half (a fill in accumulate mode; .grad = None in copy mode)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            if args.mm:  # examples/segmentation/main_MM.py:404-410: segmentation + regression objective
                logits, stage, _ = model(data)
                seg, _, _, reg = criterion(logits, data["y"], stage, 13, None, aargs)
                out["loss"] = seg + reg
            else:
                logits, stage = model(data)
                out["loss"] = criterion(logits, data["y"], stage, 13, None, aargs)
        out["loss"].backward()
        if flatg is not None:
            flatg.gather()  # copy mode: one multi-tensor copy into the all-reduce buffer, .grad -> its views

    # Software pipeline over consecutive batches.  The coordinate-only half of a step (amcontrast3d_amd/geometry.py) does
    # not depend on features or weights, so it runs ahead, on two side queues, while the current batch runs its feature half
    # on the main stream (index arithmetic in amcontrast3d_amd/schedule.py):
    #     sampling queue  FPS 24000 -> 6000 of J = --fps-lanes future batches (default 2) as ONE launch every J steps: one
    #                     workgroup per cloud, a chain of 6000 dependent iterations (8 ms), latency-bound -- more clouds per
    #                     launch cost nothing.  Lane l of a launch is consumed J + l steps later.  With J = 2 the sampling
    #                     levels 2-4 of batch t+2 (6000 -> 1500 -> 375 -> 93, 2.3 ms) run on the same queue every step, ahead
    #                     of the launch; with J > 2 (64k / 120k-point clouds) the launch runs every level itself
    #     geometry queue  neighbourhoods of batch t+1: ball queries, relative positions, reverse lists, 3-NN, and the loss
    #                     geometry (k-NN, class votes, positive masks, ambiguities, anchor lists); CU-masked; two captured
    #                     variants that fill two result sets in turn
    #     main            features of batch t: forward, loss, backward (+ all-reduce), clip + AdamW; two captured variants
    #                     that read the result set (and the input set) stream B's variant worked on one step earlier
    # Every step still does one full pass of each inside the timed region, on `--pool` rotating resident batches.  Each part is
    # its own hipGraph on its own stream: on ROCm 7.2 separate graphs on separate streams overlap, whereas branches inside ONE
    # captured graph are serialised with heavy per-node overhead (scratch/graph_conc.py: 1.6 ms vs 4.8 ms for three 0.95 ms
    # chains).
    from amcontrast3d_amd import geometry
    overlap = not args.no_overlap and not use_ddp
    prio = [int(v) for v in os.environ.get("AMC3D_STREAM_PRIO", "0,0,0,0").split(",")]  # main, fps lanes, a2, b
    main_s = torch.cuda.Stream(priority=prio[0])  # all work of this process runs on non-default streams (capture recipe)
    if os.environ.get("AMC3D_MAIN_CUS"):  # experiment: the main stream on a CU-masked queue of its own, "first:count"
        from amcontrast3d_amd import ops as _ops0
        _f, _n = (int(v) for v in os.environ["AMC3D_MAIN_CUS"].split(":"))
        main_s = _ops0.dedicated_stream(dev, _f, _n)
    main_s.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_s)
    lanes = args.fps_lanes
    if lanes <= 0:  # measure one first-level FPS and one eager feature step
        def _ms(fn, reps):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps
        t_fps = _ms(lambda: geometry.precompute_fps_levels(model, data["pos"], 0, 2), 1)
        plan0 = geometry.precompute(model, criterion.contrast_head, data, 13, None, aargs)
        data["_geometry"] = plan0
        t_feat = _ms(fwd_bwd, 2)  # eager: an upper bound of the captured feature half
        data.pop("_geometry")
        del plan0
        if flatg is None:
            opt.zero_grad(set_to_none=True)
        # two lanes on ONE queue deliver a sampling every t_fps: enough while t_fps stays below the step time (~1.25 x the
        # eager feature half).  The margin is wide on purpose: a third lane means a fourth dedicated queue, and with more
        # hardware queues than the 4 the runtime schedules natively the S step takes 17 ms instead of 8.7 (measured).
        if t_fps <= 1.5 * t_feat:
            # (until the end of round 2: two lanes, levels 2-4 as a stage of their own.)  Four batches per joint launch, all
            # four sampling levels in it (10 ms of chain every fourth step): the sampling queue is busy 2.5 instead of 6 ms per
            # step and the levels-2-4 stage with its buffers and hand-down is gone -- S 7.17 -> 6.96, L 10.8 -> 10.6,
            # S-MM 10.0 -> 9.9 ms/step (scratch/ab_lanes.sh; three lanes: 7.01)
            lanes = 4
        else:
            # long sampling chains (64k / 120k-point clouds): the joint launch runs every level of J batches once per J steps
            # and has to fit into J steps next to a busy chip (the L2-resident kernel runs 2-3x slower there): all levels
            # timed, 0.35 eager feature halves allowed per batch, at most 12 batches per launch.  (Until the end of round 2:
            # half a feature half, at most 8 -- XL-MM at 1 x 120000 points was bound by the chain: 8 batches 25.4, 12 20.9,
            # 16 22.1 ms/step; 2 x 64000 points: 5 batches 21.0, 8 20.4.)
            nlev = len(list((model.module if hasattr(model, "module") else model).encoder.encoder))
            t_all = _ms(lambda: geometry.precompute_fps_levels(model, data["pos"], 0, nlev), 1)
            lanes = int(min(12, max(3, -(-t_all // max(0.35 * t_feat, 1e-3)))))
        if rank == 0:
            print(f"bench.py: sampling chain {t_fps:.1f} ms (first level), eager feature half {t_feat:.1f} ms -> {lanes} batches per "
                  f"joint FPS launch", file=sys.stderr)
    lanes = max(1, lanes)
    # First-level FPS of TWO future batches as one launch every second step (16 workgroups instead of 8: the kernel is a
    # latency chain, more clouds cost nothing) instead of one launch per step: the sampling queue then delivers a batch
    # every t_fps / 2.  With one launch per step on one queue the step cannot be shorter than t_fps (8.1-8.3 ms alone),
    # which the feature half has reached; a queue per lane costs a fourth hardware queue (slower, measured below).
    # More than two lanes (64k / 120k-point clouds, whose first level takes several feature halves): the same with J = lanes
    # batches per launch, one launch every J steps -- XL-MM at 2 x 64000 points: 55 ms/step with a queue per lane, 40 ms so.
    joint = overlap and lanes >= 2 and not os.environ.get("AMC3D_NO_FPS_JOINT")
    nfps = 2 if joint else lanes  # distinct first-level launches (joint: the two J-batch buffers)
    # More than two lanes = clouds whose FPS levels 2-4 are long chains as well (120000 points: level 2 alone is 22 ms, longer
    # than the feature half): the joint launch then runs ALL sampling levels of its J batches, and the separate levels-2-4
    # stage of the pipeline disappears (its results went through one more set of buffers and one more step of latency).
    fps_all = joint and (lanes > 2 or bool(os.environ.get("AMC3D_FPS_ALL"))) and not os.environ.get("AMC3D_NO_FPS_ALL")
    a2_rides = joint and lanes == 2 and os.environ.get("AMC3D_A2_ON_FPS", "1") != "0"  # FPS levels 2-4 on the sampling queue (see below)
    # Hardware queues.  Ordinary HIP streams of a process share GPU_MAX_HW_QUEUES (default 4) queues round-robin, and
    # whatever shares a queue with a running FPS kernel (8 ms on 8 workgroups) waits for it; which stream that is
    # changes with every stream anybody creates (graph-internal branches, RCCL).  So the two long-latency chains get
    # queues of their own (csrc/api.hip: amc3d_stream_create_dedicated), shared only by work that is serial anyway:
    #   Q_fps : the first-level FPS of both lanes (one launch per step, 8 ms each)
    #   Q_geo : FPS levels 2-4, then the neighbourhood / loss geometry of the same batch
    from amcontrast3d_amd import ops as _ops
    # others, for scratch/queue_sweep.sh: "pooled", "fps,a2,b", ...; more than two lanes: a queue per lane (they must overlap)
    qplan = os.environ.get("AMC3D_QUEUES", "fps,geo" if (lanes <= 2 or joint) else ",".join([f"fps{l}" for l in range(lanes)] + ["geo"]))
    if (not use_graph and "AMC3D_QUEUES" not in os.environ) or qplan == "probed":
        # launched kernel by kernel, dedicated queues lose the overlap (pipeline.py): two pooled streams probed to sit
        # on hardware queues of their own, as the eager GeometryPrefetcher uses them
        from amcontrast3d_amd import pipeline as _pipeline
        s_fps_pool, s_geo_pool = _pipeline._side_streams(dev)
        s_a, s_a2, s_b = [s_fps_pool] * lanes, s_geo_pool, s_geo_pool
    elif qplan == "pooled":
        s_a = [torch.cuda.Stream(priority=prio[1]) for _ in range(lanes)]
        s_a2, s_b = torch.cuda.Stream(priority=prio[2]), torch.cuda.Stream(priority=prio[3])
    else:
        # CU masks of the side queues, "queue:first:count,...".  Default: the neighbourhood-geometry queue runs on 3/4 of the
        # CUs.  Its kernels are background work with slack (they finish ~1 ms before the feature half does), and when they
        # may spread over the whole chip they slow the feature half more than they gain: measured 8.98 -> 8.75 ms/step
        # (S, B=8 x 24000; 1/2 of the CUs: 9.16, 7/8: 8.92).  With FPS levels 2-4 moved to the sampling queue (below) the
        # searches have the whole step and 9/16 of the CUs is the best split (7/16: 8.65, 8/16 and 9/16: 8.35, 12/16: 8.47).
        # AMC3D_CU_MASK="" turns the masks off.
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        # (measured for the two-lane configuration only: with more lanes -- 64k / 120k-point clouds -- no mask by default)
        # (re-swept at the end of round 2, after the 3-NN / residual-branch changes: 9/16 7.22, 10/16 7.17, 12/16 7.24)
        # clouds of <= 24576 points (register-resident FPS kernel: the configurations the mask was measured on), any lane count
        geo_cus = (10 * ncu // 16 if (a2_rides or fps_all) else 3 * ncu // 4) if (lanes == 2 or args.points <= 24576) else 0
        cum = {k: (int(a), int(b)) for k, a, b in
               (v.split(":") for v in os.environ.get("AMC3D_CU_MASK", f"geo:0:{geo_cus}").split(",") if v)}
        q = {k: _ops.dedicated_stream(dev, *cum.get(k, (0, 0))) for k in qplan.split(",")}
        s_a = [q.get(f"fps{l}", q.get("fps")) or torch.cuda.Stream() for l in range(lanes)]
        s_a2 = q.get("a2", q.get("geo")) or torch.cuda.Stream()
        s_b = q.get("b", q.get("geo")) or torch.cuda.Stream()
    # FPS levels 2-4 (a 2.3 ms latency chain on a few workgroups) share the first level's queue and are launched ahead of
    # it: the queue is busy 2 x 2.3 + 8.3 ms in every two steps, and the geometry queue is left to the neighbourhood
    # searches, which then start 2.3 ms earlier in the step and can be held to fewer CUs (see AMC3D_CU_MASK below):
    # 8.53 -> 8.35 ms/step.  AMC3D_A2_ON_FPS=0: levels 2-4 on the geometry queue, as before.
    a2_first = False
    if a2_rides and s_a[0] is s_a[1] and s_a2 is s_b:
        s_a2, a2_first = s_a[0], True
    ev_lane = [torch.cuda.Event() for _ in range(lanes)]
    ev_a2, ev_b, ev_main = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
    head = criterion.contrast_head
    nlevels = len(list((model.module if hasattr(model, "module") else model).encoder.encoder))

    def geo_fps_first(batch):  # encoder stage 0 is the stride-1 stem (no sampling); stage 1 holds the first FPS
        return geometry.precompute_fps_levels(model, batch["pos"], 0, 2)

    def geo_fps_all(batch):
        return geometry.precompute_fps_levels(model, batch["pos"], 0, nlevels)

    def geo_fps_tail(first_level):
        return geometry.precompute_fps_levels(model, first_level[-1]["new_p"], 2, nlevels)

    def geo_rest(batch, fps):
        return geometry.precompute_rest(model, head, batch, fps, 13, None, aargs)

    def copy_batch(dst, src):
        for k in dst:
            if k != "_geometry":
                dst[k].copy_(src[k])

    # the pipeline's period: lane and pool index of step s are s % lanes and (s + lanes + 3) % npool
    import math
    from amcontrast3d_amd import schedule
    period = npool  # (set below for the overlapped pipeline)
    # Hand-over of the neighbourhood / loss geometry (~100 MB per batch) from stream B to the feature half: with graphs the
    # two never meet in a copy.  Stream B has two captured variants that write their results into two result sets R[0], R[1]
    # (the graphs' own output tensors), the feature half two variants that read them: step n reads R[n % 2] while B fills
    # R[(n + 1) % 2] for the next batch.  (Eager mode, and AMC3D_NO_PINGPONG=1, copy b_out -> cur_rest on the main stream
    # instead: 0.26 ms of multi-tensor copies per step, and as much again on stream B.)
    pingpong = overlap and use_graph and not os.environ.get("AMC3D_NO_PINGPONG")
    if overlap:
        period = schedule.period(lanes, npool, pingpong, joint)
    if overlap:
        # every in-flight batch has its own static input buffers, handed down the pipeline by rotate():
        #   in_a[lane] (first FPS level, batches t+3..) -> in_a1s (FPS levels 2-4, t+2) -> in_b (neighbourhoods, t+1) -> data (t)
        in_b = {k: v.clone() for k, v in pool[1 % npool].items()}
        in_a1s = {k: v.clone() for k, v in pool[2 % npool].items()}
        in_a = [{k: v.clone() for k, v in pool[(3 + l) % npool].items()} for l in range(lanes)]
        a1_out = [geo_fps_first(in_a[l]) for l in range(lanes)]  # written by streams A1[lane]: first FPS level
        if joint:
            # two J-batch buffers (J = lanes), launched alternately every J steps; lane l of a launch is consumed J + l steps
            # later, through per-lane views of the joint input / output tensors
            nbat = args.batch
            in_aJ = [{k: torch.cat([pool[(3 + lanes * j + t) % npool][k] for t in range(lanes)]) for k in pool[0]}
                     for j in range(2)]
            a1_outJ = [(geo_fps_all if fps_all else geo_fps_first)(in_aJ[j]) for j in range(2)]
            in_a = [[{k: v[l * nbat:(l + 1) * nbat] for k, v in in_aJ[j].items()} for l in range(lanes)] for j in range(2)]
            a1_out = [[geometry._walk(a1_outJ[j], lambda t, l=l: t[l * nbat:(l + 1) * nbat]) for l in range(lanes)]
                      for j in range(2)]
        a1_stable = geometry.clone(geo_fps_first(in_a1s))  # batch t+2: read by stream A2
        a2_out = geo_fps_tail(a1_stable)       # written by stream A2 (batch t+2): FPS levels 2..4
        fb = geo_fps_first(in_b)
        a_stable = geometry.clone(fb + geo_fps_tail(fb))  # batch t+1: read by stream B
        b_out = geometry.split(geo_rest(in_b, a_stable))[1]  # stream B (batch t+1): neighbourhoods, 3-NN, loss geometry
        fc = geo_fps_first(data)
        cur = geometry.clone(geo_rest(data, fc + geo_fps_tail(fc)))  # batch t: read by the feature half
        cur_fps, cur_rest = geometry.split(cur)
        data["_geometry"] = cur
        torch.cuda.synchronize()

    # "direct": set once the ping-pong variants are captured; "inputs": [(batch buffers, FPS picks)] x 2 once the feature
    # variants read stream B's input buffers themselves (then nothing at all moves on the main stream between steps)
    handover = {"direct": False, "inputs": None}

    def rotate(s=0):  # main stream, between steps: what the feature half of the new step reads
        if not overlap:
            copy_batch(data, pool[s % npool])
            return
        if handover["inputs"] is not None:
            return
        geometry.copy_into(cur_fps, a_stable)
        if not handover["direct"]:
            geometry.copy_into(cur_rest, b_out)
        copy_batch(data, in_b)

    def first_level(buf, what):  # what: in_a / a1_out; buf: (lane,) or (joint buffer, lane) from schedule.side_step
        for i in buf:
            what = what[i]
        return what

    def rotate_side(s=0):  # geometry queue, after rotate(): advance the side streams' buffers by one batch
        plan = schedule.side_step(s, lanes, joint, npool)
        # stream B's inputs for the batch after this one (ping-pong: the set the NEXT step's feature variant reads as well)
        tb, ta = handover["inputs"][schedule.variants(s, True)[1]] if handover["inputs"] is not None else (in_b, a_stable)
        if fps_all:  # the consumed lane holds every sampling level: straight into stream B's inputs
            geometry.copy_into(ta, first_level(plan["consume"], a1_out))
            copy_batch(tb, first_level(plan["consume"], in_a))
        else:
            geometry.copy_into(ta, a1_stable + a2_out)
            copy_batch(tb, in_a1s)
            geometry.copy_into(a1_stable, first_level(plan["consume"], a1_out))
            copy_batch(in_a1s, first_level(plan["consume"], in_a))
        for buf, pi in plan["load"]:
            copy_batch(first_level(buf, in_a), pool[pi])

    def body_a(lane=0):  # joint mode: `lane` is the index of the double-batch buffer
        if joint:
            geometry.copy_into(a1_outJ[lane], (geo_fps_all if fps_all else geo_fps_first)(in_aJ[lane]))
        else:
            geometry.copy_into(a1_out[lane], geo_fps_first(in_a[lane]))

    def body_a2():
        geometry.copy_into(a2_out, geo_fps_tail(a1_stable))

    def body_b():
        geometry.copy_into(b_out, geometry.split(geo_rest(in_b, a_stable))[1])

    fused_update = hasattr(opt, "step") and type(opt).__name__ == "FusedAdamW"

    def update():
        if fused_update:  # clip_grad_norm_(params, 10, 2) + AdamW as two launches (csrc/optim.hip)
            opt.step(max_grad_norm=10)
        else:
            torch.nn.utils.clip_grad_norm_(params, 10, norm_type=2)
            opt.step()

    step_no = [0]

    ev_rot = torch.cuda.Event()

    def run_step(f_rotate, f_side, f_a, f_a2, f_b, f_feat, f_update):
        sidx = step_no[0] % period
        step_no[0] += 1

        def critical_path():  # main stream: features (+ gradient all-reduce) + update
            (f_feat[schedule.variants(sidx, True)[0]] if isinstance(f_feat, list) else f_feat)()
            if flatg is not None:
                flatg.allreduce()
            f_update()

        if not overlap:
            f_rotate[sidx]()
            critical_path()
            return
        lane = sidx % lanes  # the FPS lane launched `lanes` steps ago delivers now and is relaunched
        # The main stream only moves what the feature half reads (the batch and its FPS picks; nothing at all once the
        # feature variants read stream B's input sets); everything else of the hand-down -- five groups of small copies between
        # the side streams' buffers -- runs on the geometry queue, off the critical path.  ev_b also orders this step's
        # rotate() after the previous step's rotate_side() (same queue).
        main_s.wait_event(ev_b)
        f_rotate[sidx]()
        ev_main.record(main_s)
        # the critical path is launched FIRST: the side launches below take the host 0.3-0.5 ms, during which the main stream
        # used to sit idle at the start of every step (HIP-event timeline: the feature graph started 0.3-0.5 ms into the step)
        launch_first = not os.environ.get("AMC3D_SIDE_FIRST")
        if launch_first:
            critical_path()
        skip = os.environ.get("AMC3D_SKIP", "")  # diagnostic: leave pipeline parts out (results go stale, timing only)
        # the first-level launch whose result is consumed now / the one (re)launched (joint: even steps only)
        plan = schedule.side_step(sidx, lanes, joint, npool)
        took, go = plan["wait"], plan["launch"]
        with torch.cuda.stream(s_b):
            s_b.wait_event(ev_main)        # rotate() has read in_b / a_stable
            s_b.wait_event(ev_lane[took])  # events, not stream waits: several parts may share a queue
            if not fps_all:
                s_b.wait_event(ev_a2)
            f_side[sidx]()
            ev_rot.record(s_b)

        def launch_fps():
            if go is not None:
                with torch.cuda.stream(s_a[go]):
                    s_a[go].wait_event(ev_rot)
                    if "fps" not in skip:
                        f_a[go]()
                    ev_lane[go].record(s_a[go])
        if not a2_first:
            launch_fps()
        if not fps_all:
            with torch.cuda.stream(s_a2):
                s_a2.wait_event(ev_rot)
                if "a2" not in skip:
                    f_a2()
                ev_a2.record(s_a2)
        if a2_first:
            launch_fps()
        with torch.cuda.stream(s_b):
            if "geo" not in skip:
                (f_b[schedule.variants(sidx, True)[1]] if isinstance(f_b, list) else f_b)()
            ev_b.record(s_b)
        if not launch_first:
            critical_path()

    def eager_step():
        if flatg is None:
            opt.zero_grad(set_to_none=True)
        run_step([lambda j=j: rotate(j) for j in range(period)], [lambda j=j: rotate_side(j) for j in range(period)],
                 [lambda l=l: body_a(l) for l in range(nfps)], body_a2,
                 body_b, fwd_bwd, update)

    step = eager_step
    if use_graph:
        # PyTorch's whole-network capture recipe, one graph per pipeline part: ~700 launches per step
        # become 6 graph launches
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        if world > 1 or sync_bn:
            # before the captures: the warm-up steps' collectives ran on a stream that never captures (graphs.on_side_stream),
            # and c10d's watchdog gets time to retire them anyway (graphs.quiesce) -- it must not poll an event of a stream
            # that is capturing
            from amcontrast3d_amd.graphs import quiesce
            quiesce()
        if flatg is None:
            opt.zero_grad(set_to_none=True)
        names = (["a2", "b", "feat", "update"] + [f"rotate{j}" for j in range(period)] + [f"side{j}" for j in range(period)]
                 + [f"fps{l}" for l in range(nfps)])
        graphs = {k: torch.cuda.CUDAGraph() for k in names}
        cap = main_s if not os.environ.get("AMC3D_CAPTURE_SIDE") else torch.cuda.Stream()
        # with a process group alive, RCCL's watchdog thread polls events while we capture: only this thread's calls
        # may be policed by the capture (the default "global" mode turns that poll into a fatal error)
        cap_mode = os.environ.get("AMC3D_CAPTURE_MODE") or ("thread_local" if world > 1 or sync_bn else "global")
        outs = []

        def capture_feat(key, pool_of=None):
            # A second variant must deliver its gradients in the FIRST variant's .grad tensors (the update graph reads
            # those): captured with .grad still set, autograd would ACCUMULATE into them -- last step's gradient plus
            # this one's.  So it runs with .grad = None and ends with one multi-tensor copy into the first variant's buffers.
            keep = [p.grad for p in params] if pool_of else None

            def body():
                if keep is not None and flatg is None:
                    for p in params:
                        p.grad = None
                fwd_bwd()
                if keep is not None and flatg is None:
                    pairs = [(g0, p.grad) for g0, p in zip(keep, params) if g0 is not None and p.grad is not None]
                    torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
            if sync_bn:
                # the SyncBatchNorm statistics all-reduces are NOT captured: the feature half becomes a chain of graphs with
                # the collectives issued eagerly between them (amcontrast3d_amd/graphs.py)
                from amcontrast3d_amd.graphs import SegmentedGraph
                graphs[key] = SegmentedGraph(cap_mode).capture(body, stream=cap)
            else:
                kw = {"pool": graphs[pool_of].pool()} if pool_of else {}  # the variants never run at the same time
                with torch.cuda.graph(graphs[key], stream=cap, capture_error_mode=cap_mode, **kw):
                    body()
            if keep is not None and flatg is None:
                assert all((g0 is None) == (p.grad is None) for g0, p in zip(keep, params)), "variants disagree on which parameters get gradients"
                for p, g0 in zip(params, keep):
                    p.grad = g0
            outs.append(out["loss"])

        if pingpong:
            graphs["b1"], graphs["feat1"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # Two sets of stream B's inputs (batch buffers + FPS picks) as well: variant v of B reads set v and fills result
            # set R[v]; one step later variant v of the feature half reads the SAME set v and R[v] -- so the batch and its picks
            # are never copied into buffers of the feature half's own, and the main stream does nothing between two steps.
            in_b2 = [in_b, {k: v.clone() for k, v in in_b.items()}]
            a_st2 = [a_stable, geometry.clone(a_stable)]
            rest = []
            for v, key in enumerate(("b", "b1")):
                with torch.cuda.graph(graphs[key], stream=s_b, capture_error_mode=cap_mode):
                    rest.append(geometry.split(geo_rest(in_b2[v], a_st2[v]))[1])  # the graph's own outputs: R[0], R[1]
            # a result set may only alias the inputs of its own variant (those belong to the same batch)
            for v, r in enumerate(rest):
                other = set()
                geometry._walk([a_st2[1 - v], in_b2[1 - v], in_a1s, a1_stable, a2_out],
                               lambda t: other.add(t.untyped_storage().data_ptr()))
                geometry._walk(r, lambda t: None if t.untyped_storage().data_ptr() not in other else
                               sys.exit("bench.py: the geometry plan aliases another batch's buffers; run with AMC3D_NO_PINGPONG=1"))
            handover["direct"] = True
            own = dict(data)  # the feature half's own buffers (the eager per-operator timing below uses them again)
            for v, key in enumerate(("feat", "feat1")):
                data.update(in_b2[v])
                data["_geometry"] = geometry.join(a_st2[v], rest[v])
                capture_feat(key, "feat" if v else None)
            data.update(own)
            handover["inputs"] = list(zip(in_b2, a_st2))
        else:
            capture_feat("feat")
        if fused_update:
            opt.prepare()  # the .grad tensors are the feature graph's now: rebuild the optimizer's tensor table before capture
        with torch.cuda.graph(graphs["update"], stream=cap, capture_error_mode=cap_mode):
            update()
        for j in range(period):
            if handover["inputs"] is None:  # (otherwise nothing moves on the main stream: no graph to replay)
                with torch.cuda.graph(graphs[f"rotate{j}"], stream=cap, capture_error_mode=cap_mode):
                    rotate(j)
            if overlap:
                with torch.cuda.graph(graphs[f"side{j}"], stream=s_b, capture_error_mode=cap_mode):
                    rotate_side(j)
        if overlap:
            for l in range(nfps):
                with torch.cuda.graph(graphs[f"fps{l}"], stream=s_a[l], capture_error_mode=cap_mode):
                    body_a(l)
            if not fps_all:
                with torch.cuda.graph(graphs["a2"], stream=s_a2, capture_error_mode=cap_mode):
                    body_a2()
            if not pingpong:
                with torch.cuda.graph(graphs["b"], stream=s_b, capture_error_mode=cap_mode):
                    body_b()
        torch.cuda.synchronize()
        if pingpong:
            # in_b / a_stable (= set 0) still hold the batch the next step works on: it goes into the set the next step's
            # feature variant reads, and its geometry into that set's results
            v0 = schedule.variants(step_no[0] % period, True)[0]
            with torch.cuda.stream(s_b):
                if v0 == 1:
                    copy_batch(in_b2[1], in_b2[0])
                    geometry.copy_into(a_st2[1], a_st2[0])
                graphs["b1" if v0 else "b"].replay()
            torch.cuda.synchronize()

        def rot_replay(j):
            return graphs[f"rotate{j}"].replay if handover["inputs"] is None else (lambda: None)

        def step():
            run_step([rot_replay(j) for j in range(period)], [graphs[f"side{j}"].replay for j in range(period)],
                     [graphs[f"fps{l}"].replay for l in range(nfps)],
                     graphs["a2"].replay, [graphs["b"].replay, graphs["b1"].replay] if pingpong else graphs["b"].replay,
                     [graphs["feat"].replay, graphs["feat1"].replay] if pingpong else graphs["feat"].replay,
                     graphs["update"].replay)

    if os.environ.get("AMC3D_CHECK_VARIANTS") and rank == 0:
        # diagnostic: total gradient norm after each of 8 steps (a variant that accumulated onto the other's gradients
        # would show as alternating norms)
        norms = []
        for _ in range(8):
            step()
            torch.cuda.synchronize()
            norms.append(round(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params if p.grad is not None))), 4))
        print("gradient norms per step:", norms, file=sys.stderr)
    if os.environ.get("AMC3D_CHECK_BATCHES") and rank == 0 and use_graph and overlap and handover["inputs"] is not None:
        # diagnostic: after every step, the set the feature half just read must hold ONE batch -- its first-level FPS picks
        # index ITS points (new_p == pos[fps_idx]), all four sampling levels chain, and the batches come round-robin from the pool
        seen = []
        drain = 4 * lanes + 8  # the initial contents of the in-flight buffers
        for n in range(3 * period + drain):
            v = schedule.variants(step_no[0] % period, True)[0]
            step()
            torch.cuda.synchronize()
            batch, fps = handover["inputs"][v]
            pts = batch["pos"]
            for lvl in fps:
                if lvl.get("fps_idx") is None:
                    continue
                want = torch.gather(pts, 1, lvl["fps_idx"].unsqueeze(-1).expand(-1, -1, 3))
                assert torch.equal(want, lvl["new_p"]), f"step {n}: the sampling plan does not belong to the batch the feature half read"
                pts = lvl["new_p"]
            seen.append(round(float(batch["pos"].double().sum()), 3))
        tail = seen[drain:]
        distinct = sorted(set(tail))
        assert len(distinct) == npool and all(tail[i] == tail[i + npool] for i in range(len(tail) - npool)), seen
        print(f"batch check: {len(seen)} steps, {len(distinct)} distinct batches round-robin, every step's sampling plan "
              f"indexes its own points (lanes {lanes}, period {period})", file=sys.stderr)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    adist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = adist.max_over_ranks(dt, dev)
    if use_graph and len(outs) == 2:  # the variant the last step replayed
        out["loss"] = outs[(step_no[0] - 1) % period % 2]
    final_loss = float(out["loss"].detach())
    assert flatg is None or flatg.intact(), "a parameter gradient left the flat all-reduce buffer"
    # data-parallel sanity: after the timed steps every rank must hold the same weights (the gradient exchange is the
    # only thing that keeps them equal: ranks see different scenes)
    replicas_in_sync = None
    if world > 1:
        import torch.distributed as tdist
        chk = torch.stack([p.detach().double().sum() for p in params] + [p.detach().double().abs().sum() for p in params])
        lo, hi = chk.clone(), chk.clone()
        from amcontrast3d_amd.graphs import on_side_stream
        on_side_stream(lambda: (tdist.all_reduce(lo, op=tdist.ReduceOp.MIN), tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)))
        replicas_in_sync = bool(torch.equal(lo, hi))

    if use_graph and overlap and rank == 0 and os.environ.get("AMC3D_TIMELINE"):
        # GPU start/end of every pipeline part relative to the step's first launch (HIP events around each replay)
        def timed(fn, stream, tag, log):
            def run():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream); fn(); e1.record(stream)
                log.append((tag, e0, e1))
            return run
        for it in range(6):
            log = []
            ref = torch.cuda.Event(enable_timing=True)
            ref.record(main_s)
            h0 = time.perf_counter()
            run_step([timed(rot_replay(j), main_s, "rotate", log) for j in range(period)],
                     [timed(graphs[f"side{j}"].replay, s_b, "side", log) for j in range(period)],
                     [timed(graphs[f"fps{l}"].replay, s_a[l], f"fps{l}", log) for l in range(nfps)],
                     timed(graphs["a2"].replay if not fps_all else (lambda: None), s_a2, "a2", log),
                     # ping-pong: the variants in the order run_step picks them (a feature variant must never run beside
                     # the B variant that writes the result set it reads)
                     [timed(graphs[k].replay, s_b, "b", log) for k in (("b", "b1") if pingpong else ("b", "b"))],
                     [timed(graphs[k].replay, main_s, "feat", log) for k in (("feat", "feat1") if pingpong else ("feat", "feat"))],
                     timed(graphs["update"].replay, main_s, "update", log))
            h1 = time.perf_counter()
            if it >= 3:
                torch.cuda.synchronize()
                print(f"timeline step {it}: host launch {1e3*(h1-h0):.2f} ms | " + " | ".join(
                    f"{tag} {ref.elapsed_time(e0):.2f}-{ref.elapsed_time(e1):.2f}" for tag, e0, e1 in log), file=sys.stderr)
        torch.cuda.synchronize()

    if args.lean:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "loss": final_loss, "lean": True}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    parts = None
    no_overlap_ms = None
    if use_graph and overlap:  # on every rank: with SyncBN the feature replay contains collectives
        # each pipeline part alone (back-to-back replays on its stream): what the overlap has to hide
        def alone(fn, stream, reps=5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t) / reps * 1e3, 3)
        parts = {"features_ms": alone(graphs["feat"].replay, main_s), "update_ms": alone(graphs["update"].replay, main_s),
                 "fps_level1_ms": alone(graphs["fps0"].replay, s_a[0]), "fps_levels2to4_ms": alone(graphs["a2"].replay, s_a2) if not fps_all else None,
                 "neighbourhood_geometry_ms": alone(graphs["b"].replay, s_b),
                 "rotate_ms": alone(rot_replay(0), main_s),
                 "rotate_side_ms": alone(graphs["side0"].replay, s_b)}

        # the same step with nothing overlapped: every part replayed on the stream it was captured on, one after the other
        # (a host wait between parts: ~6 x 20 us of the figure)
        def serial(reps=5):
            per = {}
            for r in range(-1, reps):  # pass -1: untimed (first replays after the per-part measurements above)
                if r == 0:
                    torch.cuda.synchronize()
                    per = {}
                    t = time.perf_counter()
                for tag, fn, st in (("rotate", rot_replay(r % period), main_s),
                                    ("rotate_side", graphs[f"side{r % period}"].replay, s_b),
                                    ("fps1", graphs[f"fps{(r // lanes) % 2 if joint else r % lanes}"].replay
                                     if not (joint and r % lanes) else (lambda: None), s_a[r % lanes]),
                                    ("fps2to4", graphs["a2"].replay if not fps_all else (lambda: None), s_a2),
                                    ("geometry", graphs["b"].replay, s_b),
                                    ("features", graphs["feat"].replay, main_s)):
                    h = time.perf_counter()
                    with torch.cuda.stream(st):
                        fn()
                    st.synchronize()
                    per[tag] = per.get(tag, 0.0) + (time.perf_counter() - h) / reps * 1e3
                if flatg is not None:
                    flatg.allreduce()
                graphs["update"].replay()
                main_s.synchronize()
            if os.environ.get("AMC3D_SERIAL_PARTS"):
                print("serial parts (ms): " + json.dumps({k: round(v, 3) for k, v in per.items()}), file=sys.stderr)
            return round((time.perf_counter() - t) / reps * 1e3, 3)
        no_overlap_ms = serial()

    # per-operator HIP-event timing: the same step, launched eagerly so each C-ABI launch can be
    # bracketed by events on its stream (events cannot bracket nodes inside a graph replay)
    ksteps = min(args.steps, 3)
    data.pop("_geometry", None)  # geometry inline on the main stream: events then bracket one launch each
    overlap_was, overlap = overlap, False
    timing.enable(True)
    for _ in range(ksteps):
        eager_step()
    torch.cuda.synchronize()
    kernels = timing.collect()
    timing.enable(False)
    for v in kernels.values():
        v["total_ms"] *= args.steps / ksteps
        v["launches"] *= args.steps / ksteps
        v["bytes"] *= args.steps / ksteps

    if rank == 0:
        points_per_step = args.batch * args.points
        ms_per_step = dt / args.steps * 1e3
        value = points_per_step * world / (dt / args.steps)
        # HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 --pmc in
        # separate runs, as gpurun requires; None for operators that were not measured)
        traffic = {}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as fh:
                traffic = json.load(fh)
        except OSError:
            pass

        def kernel_roofline(name, v):
            """bytes (or FLOPs) of all launches of the operator / their summed HIP-event time, against the roof its
            arithmetic intensity puts it under (machine balance 157.3 TF / 8 TB/s ~ 20 flop/byte)"""
            nbytes, fl, tms = v["bytes"], v["flops"] * (args.steps / ksteps), v["total_ms"]
            per_launch = {"avg_launch_ms": round(v["avg_ms"], 4), "launches_per_step": v["launches"] / args.steps,
                          "ms_per_step": round(tms / args.steps, 4),
                          "traffic": traffic.get(name, {}).get("bytes_per_launch")}
            if fl > 0 and fl / max(nbytes, 1) > F32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
                ach = fl / (tms * 1e-3) / 1e12
                return dict({"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                             "algorithmic_flops_per_launch": fl / max(v["launches"], 1)}, **per_launch)
            ach = nbytes / (tms * 1e-3) / 1e9
            return dict({"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(v["bytes_per_launch"])},
                        **per_launch)

        # the dominant kernel of the critical path: with the pipeline on, the geometry operators (FPS chain, searches,
        # loss geometry) run one to four steps ahead on their own queues and the feature half bounds the step
        crit = {k: v for k, v in kernels.items() if not (overlap_was and k in GEOMETRY_OPS)}
        roofline = None
        if crit:
            dom_name, dom = max(crit.items(), key=lambda kv: kv[1]["total_ms"])
            roofline = kernel_roofline(dom_name, dom)
            roofline["note"] = ("largest HIP-event time among the operators of the feature half (the stream that bounds "
                                "the overlapped step); the FPS chain is reported under latency_chain")
        hbm_names = [k for k in kernels if k.startswith("bn_") or k in ("group_points", "group_points_grad",
                     "three_interpolate", "three_interpolate_grad", "contrast_forward", "contrast_backward")]
        roofline_hbm = kernel_roofline(*max(((k, kernels[k]) for k in hbm_names), key=lambda kv: kv[1]["total_ms"])) \
            if hbm_names else None
        # the dense contractions together (fused grouped conv + pointwise conv + SetAbstraction tail, fp32 MFMA)
        roofline_mfma = None
        mf = [k for k in kernels if k.startswith(("grouped_conv", "pointwise_conv", "sa_tail", "local_aggregation"))]
        if mf:
            fl = sum(kernels[k]["flops"] for k in mf) * (args.steps / ksteps)
            tm = sum(kernels[k]["total_ms"] for k in mf)
            ach = fl / (tm * 1e-3) / 1e12
            roofline_mfma = {"bound": "mfma", "kernel": "+".join(sorted(mf)), "achieved": round(ach, 2),
                             "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                             "traffic": None, "ms_per_step": round(tm / args.steps, 3),
                             "note": "launched FLOPs (recomputation included) of every MFMA operator / their summed time"}
        # the FPS chain: one workgroup per cloud, n/4 dependent iterations -- a latency figure, not a bandwidth one
        latency_chain = None
        if "furthest_point_sampling" in kernels:
            lvl1 = parts["fps_level1_ms"] if parts else None
            fv = kernels["furthest_point_sampling"]
            its = args.points // 4
            # all-levels joint launch: the replayed graph is the whole chain of a cloud (n/4 + n/16 + ... iterations)
            its_timed = sum(args.points // 4 ** k for k in range(1, nlevels)) if fps_all else its
            latency_chain = {"kernel": "furthest_point_sampling", "level1_iterations": its,
                             ("all_levels_ms" if fps_all else "level1_ms"): lvl1, "iterations_timed": its_timed,
                             "us_per_iteration": round(lvl1 * 1e3 / its_timed, 3) if lvl1 else None,
                             "all_levels_ms_per_step": round(fv["total_ms"] / args.steps, 3),
                             "cus_busy": args.batch, "hbm_bytes_per_launch": traffic.get("furthest_point_sampling", {}).get("bytes_per_launch"),
                             "note": (f"serial arg-max chain, off the critical path: all sampling levels of {lanes} future batches run as one "
                                      f"launch every {lanes} steps on the sampling queue" if fps_all else
                                      "serial arg-max chain, off the critical path: the first level of two future batches runs as one launch "
                                      "every second step on the sampling queue, levels 2-4 every step ahead of it")}
        # the whole step against both roofs (SURVEY 8(d) algorithmic work)
        roofline_step = None
        if args.variant in ALGORITHMIC_PER_POINT and not args.mm:
            bpp, fpp = ALGORITHMIC_PER_POINT[args.variant]
            ab, af = bpp * points_per_step, fpp * points_per_step
            meas = traffic.get("_step", {}).get("bytes_per_step") if (args.variant == "S" and args.batch == 8 and args.points == 24000) else None
            roofline_step = {"algorithmic_GB": round(ab / 1e9, 3), "algorithmic_GFLOP": round(af / 1e9, 1),
                             "ms": round(ms_per_step, 3),
                             "hbm_GBps": round(ab / (ms_per_step * 1e-3) / 1e9, 1),
                             "hbm_frac": round(ab / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "mfma_TFLOPs": round(af / (ms_per_step * 1e-3) / 1e12, 2),
                             "mfma_frac": round(af / (ms_per_step * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                             "measured_hbm_GB": round(meas / 1e9, 2) if meas else None,
                             "algorithmic_over_measured": round(ab / meas, 3) if meas else None,
                             "note": "algorithmic = SURVEY 8(d) (ideal fusion, forward x 3); measured = sum of FETCH_SIZE x 2 + "
                                     "WRITE_SIZE over every kernel of one step (profiles/, rocprofv3 --pmc passes)"}
        line = {
            "metric": "train-step points/sec (fwd+bwd) on 24k-pt S3DIS clouds",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 (1x1 convs on the bf16 MFMA, fp32 accumulate; tensors fp32)",
            "data": "synthetic",
            "config": {"workload": f"PointNeXt-{args.variant} + AMContrast3D-{'MM (++)' if args.mm else 'AA'}, S3DIS-shaped {args.points}-pt "
                                   f"voxelised (0.04 m) clouds, batch {args.batch}/GPU, fwd + CE/contrast loss + bwd + "
                                   f"clip + AdamW",
                       "global_batch": args.batch * world, "points": args.points,
                       "parallelism": f"dp{world}" + ("+syncbn+ddp" if use_ddp else "+syncbn" if sync_bn else ""),
                       "launch": "hipGraph replay (fwd+loss+bwd | clip+AdamW)" if use_graph else "eager",
                       "pipeline": ((f"3 queues: sampling (all FPS levels of {lanes} future batches as one launch every {lanes} steps) | "
                                     "neighbourhood + loss geometry of batch t+1 (CU-masked) | features of batch t; geometry handed "
                                     "over without copies (two captured variants each)")
                                    if (joint and pingpong and fps_all) else
                                    ("3 queues: sampling (FPS levels 2-4 of batch t+2 every step, then level 1 of batches t+3 and "
                                     "t+4 as one launch every second step) | neighbourhood + loss geometry of batch t+1 (CU-masked) | "
                                     "features of batch t; geometry handed over without copies (two captured variants each)")
                                    if (joint and pingpong) else
                                    f"{3 + lanes} streams: FPS level 1 of {lanes} future batches in flight | FPS levels 2-4 "
                                    f"(t+2) | neighbourhood + loss geometry (t+1) | features (t)")
                                   if overlap_was else "none"},
            "loss": round(final_loss, 6),
            "replicas_in_sync": replicas_in_sync,
            "ms_per_step_no_overlap": no_overlap_ms,
            # one batch through every stage with nothing else on the chip: its sampling chain (all levels; a joint launch takes
            # as long for J clouds as for one), its neighbourhood / loss geometry, its feature half and update
            "single_batch_latency_ms": (round(sum(v for v in parts.values() if v), 3) if parts else None),
            "resident_batches": npool,
            "roofline": roofline,
            "roofline_step": roofline_step,
            "latency_chain": latency_chain,
            "roofline_hbm": roofline_hbm,
            "roofline_mfma": roofline_mfma,
            "kernels": {k: {"ms_per_step": round(v["total_ms"] / args.steps, 4),
                            "launches_per_step": v["launches"] / args.steps} for k, v in kernels.items()},
            "native_ms_per_step": round(sum(v["total_ms"] for v in kernels.values()) / args.steps, 3),
            "pipeline_parts_alone": parts,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, model, nb, configs.ambiguity_args("s3dis"), args.points,
                                                small=args.cpu_baseline_batch)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
