#!/bin/bash
# One-GPU-box rehearsal of bench.py --sync-bn: (1) a one-rank RCCL group with the all-reduces inside the captured
# step, (2) two ranks sharing the card over gloo, eager (gloo cannot be captured; RCCL refuses two ranks per device).
set -e
AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --sync-bn --steps 10 --warmup 3 --no-cpu-baseline \
    > gpurun_out/syncbn_graph_1rank.log 2> gpurun_out/syncbn_graph_1rank.err
tail -1 gpurun_out/syncbn_graph_1rank.log | cut -c1-300
AMC3D_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 --sync-bn --no-graph \
    --no-cpu-baseline > gpurun_out/syncbn_gloo_2rank.log 2> gpurun_out/syncbn_gloo_2rank.err
tail -1 gpurun_out/syncbn_gloo_2rank.log | cut -c1-300
