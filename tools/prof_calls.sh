#!/bin/bash
# tools/prof_calls.sh <pattern>: launches of matching kernels in one eager step (no graph, no overlap), with grid sizes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_calls
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_calls -o r -- python3 bench.py --lean --steps 3 --warmup 2 --eager > /dev/null 2> gpurun_out/prof_calls.err
for p in "$@"; do python3 tools/kernel_calls.py /tmp/prof_calls/r_results.db "$p" | grep -v columns; done
