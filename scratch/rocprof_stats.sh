#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command -> gpurun_out/<tag>_rocprof_kernel_stats.csv (top of it printed)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/rp_stats
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_stats -o r --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_rocprof_bench.json 2> gpurun_out/${TAG}_rocprof_bench.err
ls /tmp/rp_stats
f=$(ls /tmp/rp_stats/*kernel_stats.csv | head -1)
head -40 "$f" > gpurun_out/${TAG}_rocprof_kernel_stats.csv
head -12 "$f" | cut -c1-160
tail -1 gpurun_out/${TAG}_rocprof_bench.json | cut -c1-160
