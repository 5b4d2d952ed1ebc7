#!/bin/bash
# the round's evidence in one call: steady-state kernel tables (S, L), the HBM counter passes, the driver's bench command
set -e
TAG=$1
cd $GRAFT_REPO_ROOT
bash scratch/prof_steady.sh ${TAG}S > /dev/null
bash scratch/prof_steady.sh ${TAG}L --variant L > /dev/null
bash scratch/pmc_round2.sh ${TAG}S | tail -3
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-200
