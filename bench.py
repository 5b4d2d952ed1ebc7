#!/usr/bin/env python
"""Train-step throughput of the AMContrast3D hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): train-step points/sec -- forward + CrossEntropyAce (cross-entropy + adaptive-margin contrast over 4
decoder stages) + backward (+ gradient all-reduce for N > 1) + clip + AdamW step -- on synthetic S3DIS-shaped 24 000-point
clouds.  Workload at every N: BASELINE config 2, PointNeXt-S + AMContrast3D-AA, batch 8 clouds per GPU (weak scaling: scenes
are sharded across ranks, one process per GPU, SyncBN + one gradient all-reduce over RCCL as main_AA.py:146-152 asks).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

The step that is timed is the PRODUCT's: amcontrast3d_amd.pipeline.GraphPipeline (hipGraph replays on three hardware queues,
what amcontrast3d_amd.train.train_one_epoch runs); this file only builds the model, feeds resident batches and measures.
`--gpus N` without a torchrun environment starts the N ranks itself (a torch.distributed.run child process, launched before
this process has made any GPU call).  Besides the contract fields the line carries

  roofline        the dominant operator of the feature half (amcontrast3d_amd/roofline.py): SURVEY 8(d) algorithmic bytes (or
                  FLOPs) per launch / live HIP-event time vs 8 TB/s (157.3 TF fp32 MFMA); `traffic` = PMC bytes per launch
                  (profiles/hbm_traffic.json), `rocprof` = the committed per-kernel durations of the same operator
  roofline_step   the whole step against both roofs; measured HBM bytes per step and their ratio to the algorithmic bytes
  latency_chain   the FPS chain (runs J..2J steps ahead on a queue of its own): microseconds per dependent iteration
  train_one_epoch_ms_per_step   the same step through train.train_one_epoch (loader batches, confusion matrix, schedule)
  ms_per_step_no_overlap, single_batch_latency_ms, pipeline_parts_alone   the parts back to back / one batch through all
  cpu_baseline    the oracle's CPU restatement of the SAME step on this box's host cores (rank 0, N = 1 only)
  kernels         per-operator HIP-event totals of an eager step (ms per step)
"""
import argparse
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=24000)
    ap.add_argument("--variant", default="S")
    ap.add_argument("--mm", action="store_true", help="AMContrast3D++ (BaseSeg_M_AMContrast3D + CrossEntropyAcePre)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="kernel-by-kernel launches, geometry in line, no pipeline: what the rocprofv3 --pmc passes trace (every "
                         "dispatch is one kernel)")
    ap.add_argument("--no-graph", action="store_true", help="= --eager (older command lines)")
    ap.add_argument("--no-overlap", action="store_true", help="= --eager (older command lines)")
    ap.add_argument("--fps-lanes", type=int, default=0, help="batches per joint FPS launch (0 = let GraphPipeline measure and choose)")
    ap.add_argument("--no-sync-bn", action="store_true",
                    help="N > 1: per-rank BatchNorm statistics.  Default at N > 1 is the reference's behaviour (main_AA.py:146-148, "
                         "820): statistics over all ranks, exchanged between captured graph segments (amcontrast3d_amd/graphs.py)")
    ap.add_argument("--pool", type=int, default=4, help="distinct resident batches fed round-robin (different geometry every step)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="bf16: the reference's use_amp (main_AA.py:389-394)")
    ap.add_argument("--lean", action="store_true", help="profiling runs: stop after the timed loop")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: only the multi-rank control flow on a small torch CPU model over gloo (tests/test_dist_cpu.py)")
    ap.add_argument("--ddp", action="store_true", help="N > 1: torch's SyncBatchNorm + DistributedDataParallel wrappers, eager")
    ap.add_argument("--eval", action="store_true", help="whole-room testing (amcontrast3d_amd.evaluate) instead of the train step")
    ap.add_argument("--room-points", type=int, default=300000)
    ap.add_argument("--cpu-baseline-batch", type=int, default=2, help="clouds in the small CPU sample")
    a = ap.parse_args()
    a.eager = a.eager or a.no_graph or a.no_overlap
    return a


def build(variant, dev, world, ddp, mm=False, sync_bn=False):
    import amcontrast3d_amd
    amcontrast3d_amd.activate()
    from amcontrast3d_amd import configs, dist as adist
    from openpoints.loss import build_criterion_from_cfg
    from openpoints.models import build_model_from_cfg
    from openpoints.optim import build_optimizer_from_cfg
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
    opt = build_optimizer_from_cfg(model, NAME="adamw", lr=0.01, weight_decay=1e-4)  # FusedAdamW on the GPU
    return cfg, model, criterion, aargs, opt


def cpu_baseline(cfg, model, batch_np, aargs_dict, points, small=2):
    """The oracle's CPU restatement of the step on this box's host cores: 1 warm-up + 3 timed steps at
    `small` clouds (median; forward / loss / backward split), then 2 timed steps of the full batch, whose median is `value`
    (the same workload as the GPU line; ~1 minute of CPU work in all)."""
    import statistics
    from oracle import model_ref, pointops_ref
    pointops_ref.build()
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMC3D_CPU_THREADS", "16")))  # the box's CPU share for one GPU
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

    def say(ms, n, what):
        return (f"{what}, batch {n} x {points} points: {ms['step']:.2f} s/step (forward {ms['forward']:.2f}, loss {ms['loss']:.2f}, "
                f"backward {ms['backward']:.2f})")
    full = batch_np["pos"].shape[0]
    small = min(small, full)
    ms = run(small, 3, 1)
    out = {"value": small * points / ms["step"], "unit": "points/s", "cores": cores, "kind": "port",
           "sample": say(ms, small, "median of 3 steps after 1 warm-up") +
                     f"; oracle/model_ref.py + pointops_ref.c, OpenMP/torch {cores} threads; no optimizer step"}
    if full > small and not os.environ.get("AMC3D_CPU_BASELINE_SMALL_ONLY"):
        mf = run(full, 2, 0)
        out["value_small_batch"], out["value"] = out["value"], full * points / mf["step"]
        out["sample"] = say(mf, full, "median of 2 steps (the GPU line's workload)") + "; and " + out["sample"]
    return out


def launch_ranks(args):
    """`--gpus N` without a torchrun environment: start the N ranks as a CHILD process tree (torch.distributed.run) and pass
    their output through.  This process has made no GPU call (importing torch makes none)."""
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


def replicas_in_sync(params):
    """after the timed steps every rank must hold the same weights (the gradient exchange is the only thing that keeps them
    equal: ranks see different scenes): per-parameter checksums MIN- and MAX-reduced must agree bit for bit"""
    import torch.distributed as tdist
    from amcontrast3d_amd.graphs import on_side_stream
    chk = torch.stack([p.detach().double().sum() for p in params] + [p.detach().double().abs().sum() for p in params])
    lo, hi = chk.clone(), chk.clone()
    on_side_stream(lambda: (tdist.all_reduce(lo, op=tdist.ReduceOp.MIN), tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)))
    return bool(torch.equal(lo, hi))


def rehearse_cpu(args):
    """The N-rank control flow of main() on a stand-in torch CPU model over gloo (no kernels): that `--gpus N` starts N ranks,
    shards scenes, keeps replicas in sync through the flat gradient all-reduce and reports max-over-ranks time on rank 0."""
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
    sync = replicas_in_sync(params) if world > 1 else True
    if rank == 0:
        print(json.dumps({"metric": "rehearsal", "value": 0.0, "unit": "none", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3),
                          "replicas_in_sync": sync, "scene_ids_rank0": ids, "data": "synthetic"}))
    if world > 1:
        torch.distributed.destroy_process_group()


def eval_main(args):
    """one room = voxel partition into sub-clouds, eval-mode model on every sub-cloud, mean vote
    per point, whole / boundary / inner confusion matrices.  A 'step' is one whole room."""
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
    one_room = lambda: evaluate.test_cloud_boundary_inner(model, coord, feat, label, parts, 13, None, 24)  # noqa: E731
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
            "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"PointNeXt-{args.variant}, one room of {args.room_points} points in {len(parts)} sub-clouds of "
                                   f"{len(parts[0])} points (voxel 0.04), inputs on the host"},
            "miou_whole_boundary_inner": [round(v, 3) for v in evaluate.summarize(r["cm"], r["cm_b"], r["cm_i"])[0:15:5]]}
    if not args.no_cpu_baseline:
        from oracle import eval_ref, pointops_ref
        pointops_ref.build()
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("AMC3D_CPU_THREADS", "16")))
        pointops_ref.set_threads(cores); torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        t0 = time.perf_counter()
        eval_ref.test_cloud(sd, json.loads(json.dumps(configs.model_cfg(args.variant, dropout=0))), coord, feat, label_np, parts[:2], 13, None, 24)
        dtc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(2 * len(parts[0]) / dtc, 1), "unit": "points/s", "cores": cores, "kind": "port",
                                "sample": f"2 of the {len(parts)} sub-clouds, {dtc:.1f} s (oracle/eval_ref.py, {cores} threads)"}
    print(json.dumps(line))


def loader_batches(pool_np, n):
    """n batches in the reference's collated layout (point-major feature keys on the host side of the loader; here resident on
    the device already, as the bench's inputs are): what train.train_one_epoch consumes"""
    import numpy as np
    out = []
    for b in pool_np:
        out.append({"pos": torch.from_numpy(b["pos"]).cuda(), "y": torch.from_numpy(b["y"]).cuda(),
                    "x": torch.from_numpy(np.ascontiguousarray(b["x"][:, :3].transpose(0, 2, 1))).cuda(),
                    "heights": torch.from_numpy(np.ascontiguousarray(b["x"][:, 3:4].transpose(0, 2, 1))).cuda()})
    return [dict(out[i % len(out)]) for i in range(n)]  # (the loop replaces 'x' by the assembled features: a fresh dict per batch)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # children do the work; this process never touches the GPU
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    if args.eval:
        return eval_main(args)
    from amcontrast3d_amd import _lib, configs, dist as adist, roofline, synthetic, timing
    from amcontrast3d_amd.pipeline import GraphPipeline
    rank, local, world = adist.init_from_env()
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
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        blocks._FORCE_SYNCED_BN = sync_bn = True
    eager = args.eager or use_ddp
    cfg, model, criterion, aargs, opt = build(args.variant, dev, world, use_ddp, args.mm, sync_bn)
    # `--pool` distinct resident batches rotate through the pipeline: every step sees another cloud geometry (k-NN tie counts,
    # grid occupancy, cache contents), as a training loop would; scene ids are disjoint across ranks and steps
    pool_np = [synthetic.make_batch(args.batch, args.points, first_id=adist.scene_ids(rank, world, args.batch, step=j)[0])
               for j in range(max(1, args.pool))]
    pool = [{k: torch.from_numpy(v).to(dev) for k, v in b.items()} for b in pool_np]
    params = list(model.parameters())
    flatg = (adist.FlatGradients(params, accumulate=False) if (world > 1 and not use_ddp) or os.environ.get("AMC3D_FLAT_GRADS") else None)
    amp = torch.bfloat16 if args.dtype == "bf16" else None

    def step_loss(data):
        if args.mm:  # examples/segmentation/main_MM.py:404-410: segmentation + regression objective
            logits, stage, _ = model(data)
            seg, _, _, reg = criterion(logits, data["y"], stage, 13, None, aargs)
            return logits, seg + reg, ()
        logits, stage = model(data)
        return logits, criterion(logits, data["y"], stage, 13, None, aargs), ()

    def eager_step(data):  # kernel by kernel, geometry in line: the PMC passes and the per-operator timing
        opt.zero_grad(set_to_none=True) if flatg is None else flatg.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            logits, loss, _ = step_loss(data)
        loss.backward()
        if flatg is not None:
            flatg.gather()
            flatg.allreduce()
        opt.step(max_grad_norm=10) if type(opt).__name__ == "FusedAdamW" else (torch.nn.utils.clip_grad_norm_(params, 10), opt.step())
        return loss

    main_s = torch.cuda.Stream()  # all work of this process runs on a non-default stream (capture recipe)
    main_s.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_s)
    pipe = None
    if eager:
        feed = itertools.cycle(pool)
        step = lambda: eager_step(dict(next(feed)))  # noqa: E731
    else:
        def make_pipeline():
            return GraphPipeline(model, step_loss, criterion.contrast_head, opt, pool[0], 13, None, aargs, lanes=args.fps_lanes,
                                 max_grad_norm=10, flat_grads=flatg, sync_bn=sync_bn, keep_state=False, amp_dtype=amp,
                                 verbose=rank == 0)
        try:
            pipe = make_pipeline()
        except Exception as e:  # noqa: BLE001
            # RCCL collectives recorded into the graphs (round 3) were rehearsed with a one-rank group only: should the capture be
            # refused on a real node (it fails alike on every rank), fall back to round 2's form -- graphs cut at the collectives
            if world == 1 and not sync_bn or os.environ.get("AMC3D_SEGMENTED_COLLECTIVES"):
                raise
            print(f"bench.py: capturing the collectives failed ({type(e).__name__}: {str(e)[:200]}); retrying with graph segments",
                  file=sys.stderr)
            os.environ["AMC3D_SEGMENTED_COLLECTIVES"] = "1"
            torch.cuda.synchronize()
            pipe = make_pipeline()
        runner = pipe.run(itertools.cycle(pool))
        step = lambda: next(runner)["loss"]  # noqa: E731
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    adist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    adist.barrier()
    torch.cuda.synchronize()
    dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
    final_loss = float(loss.detach())
    assert flatg is None or flatg.intact(), "a parameter gradient left the flat all-reduce buffer"
    in_sync = replicas_in_sync(params) if world > 1 else None
    ms_per_step = dt / args.steps * 1e3
    if args.lean:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(ms_per_step, 3), "loss": final_loss, "lean": True}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    parts = serial_ms = epoch_ms = None
    if pipe is not None:  # on every rank: with SyncBN the feature replay contains collectives
        parts, serial_ms = pipe.parts_alone(), pipe.serial_ms()
        if world == 1 and not args.mm and amp is None:
            # the same step through the product's training loop (main_AA.py:370-428): loader batches in the reference's collated
            # layout, confusion matrix, loss averaging, cosine schedule stepped per iteration -- on the pipeline built above
            from amcontrast3d_amd import train
            from openpoints.scheduler import build_scheduler_from_cfg
            from openpoints.utils import EasyConfig
            tcfg = EasyConfig()
            tcfg.update({"num_classes": 13, "ignore_index": None, "ambiguity_args": aargs, "feature_keys": "x,heights", "use_amp": False,
                         "step_per_update": 1, "grad_norm_clip": 10, "sched_on_epoch": False, "fps_lanes": pipe.lanes})
            scfg = EasyConfig()
            scfg.update({"sched": "cosine", "epochs": 100, "min_lr": 1e-5, "warmup_epochs": 0, "lr": 0.01})
            sched = build_scheduler_from_cfg(scfg, opt)
            nb = max(200, 5 * args.steps)  # an S3DIS-sized epoch (~200 iterations): filling the pipeline once (one joint FPS launch + one geometry pass, ~14 ms) is then 1 % of it
            for rep in range(2):  # the first epoch builds (and caches) the loop's own pipeline
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                train.train_one_epoch(model, loader_batches(pool_np, nb), criterion, opt, sched, None, 1 + rep, tcfg)
                torch.cuda.synchronize()
                epoch_ms = (time.perf_counter() - t0) / nb * 1e3
    # per-operator HIP-event timing: the same step launched eagerly so that each C-ABI launch can be bracketed by events on its
    # stream (events cannot bracket nodes inside a graph replay)
    ksteps = min(args.steps, 3)
    timing.enable(True)
    for j in range(ksteps):
        eager_step(dict(pool[j % len(pool)]))
    torch.cuda.synchronize()
    kernels = timing.collect()
    timing.enable(False)
    if rank == 0:
        fps = None
        if pipe is not None:
            fps = {"points": args.points, "levels": pipe.nlevels, "joint_ms": parts["fps_all_levels_joint_launch_ms"],
                   "clouds": args.batch * pipe.lanes, "lanes": pipe.lanes}
        roofs = roofline.report(kernels, ksteps, ms_per_step, args.batch * args.points, args.variant, args.mm, pipe is not None, fps)
        line = {
            "metric": "train-step points/sec (fwd+bwd) on 24k-pt S3DIS clouds",
            "value": round(args.batch * args.points * world / (dt / args.steps), 1), "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if amp is None else "bf16 (autocast: 1x1 convs on the bf16 MFMA, fp32 accumulate and statistics)",
            "data": "synthetic",
            "config": dict({"workload": f"PointNeXt-{args.variant} + AMContrast3D-{'MM (++)' if args.mm else 'AA'}, S3DIS-shaped "
                                        f"{args.points}-pt voxelised (0.04 m) clouds, batch {args.batch}/GPU, fwd + CE/contrast loss + "
                                        f"bwd + clip + AdamW",
                            "global_batch": args.batch * world, "points": args.points,
                            "parallelism": f"dp{world}" + ("+syncbn+ddp" if use_ddp else "+syncbn" if sync_bn else "")},
                           **(pipe.describe() if pipe is not None else {"launch": "eager", "pipeline": "none"})),
            "loss": round(final_loss, 6), "replicas_in_sync": in_sync, "resident_batches": len(pool),
            "train_one_epoch_ms_per_step": round(epoch_ms, 3) if epoch_ms else None,
            "ms_per_step_no_overlap": serial_ms,
            "single_batch_latency_ms": round(sum(v for v in parts.values() if v), 3) if parts else None,
            "pipeline_parts_alone": parts,
            **roofs,
            "kernels": {k: {"ms_per_step": round(v["total_ms"] / ksteps, 4), "launches_per_step": v["launches"] / ksteps}
                        for k, v in kernels.items()},
            "native_ms_per_step": round(sum(v["total_ms"] for v in kernels.values()) / ksteps, 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, model, pool_np[0], configs.ambiguity_args("s3dis"), args.points,
                                                small=args.cpu_baseline_batch)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
