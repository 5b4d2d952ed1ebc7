#!/bin/bash
# steady-state kernel table of bench.py: scratch/prof_steady.sh <tag> [bench args...]  -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o r -- python3 bench.py --lean --steps 8 --warmup 3 "$@" > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err
python3 scratch/steady2.py /tmp/prof_$TAG/r_results.db gpurun_out/${TAG}_kernel_stats.csv 5 > gpurun_out/${TAG}_steady.txt
head -70 gpurun_out/${TAG}_steady.txt
