#!/bin/bash
# HBM traffic of every kernel of the train step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md: they do not fit one pass), summarised on the box (the raw databases are too large to merge back).
# Eager launches (--no-graph --no-overlap) so that every dispatch is attributed to its kernel.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o r -- python3 bench.py --steps 3 --warmup 2 --no-graph --no-overlap --no-cpu-baseline > gpurun_out/pmc_$C.log 2> gpurun_out/pmc_$C.err
  python3 scratch/pmc_extract.py /tmp/pmc_$C/r_results.db $C gpurun_out/pmc_$C.csv | tail -8
done
