#!/bin/bash
# steady-state kernel table of the pipelined step: tools/prof_steady.sh <tag> [bench args...]
#   -> gpurun_out/<tag>_kernel_stats.csv (per kernel and stream: calls / us per step, average / min / max us), gpurun_out/<tag>_steady.txt
# rocprofv3 --kernel-trace of `bench.py --lean` (the process ends right after the timed loop); the window is the last 5 steps.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o r -- python3 bench.py --lean --steps 12 --warmup 12 "$@" > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err
python3 tools/steady2.py /tmp/prof_$TAG/r_results.db gpurun_out/${TAG}_kernel_stats.csv 5 > gpurun_out/${TAG}_steady.txt
head -${LINES_SHOWN:-70} gpurun_out/${TAG}_steady.txt
