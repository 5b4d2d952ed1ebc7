#!/bin/bash
# scratch/prof_calls_xl.sh <pattern>...: as prof_calls.sh, for XL-MM at 2 x 64000 points
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_calls
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_calls -o r -- python3 bench.py --lean --steps 3 --warmup 2 --no-graph --no-overlap --fps-lanes 2 --variant XL --mm --batch 2 --points 64000 > /dev/null 2> gpurun_out/prof_calls.err
for p in "$@"; do python3 scratch/kernel_calls.py /tmp/prof_calls/r_results.db "$p" | grep -v columns; done
