#!/bin/bash
# HBM traffic of every kernel of the train step: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md:
# they do not fit one pass; --pmc only with --kernel-trace), summarised on the box.  Eager launches (bench.py --eager --lean) so that
# every dispatch is one kernel; the run makes 7 train steps, the last 3 are summed (tools/pmc_extract.py), FETCH doubled
# (tools/pmc_merge.py).       usage: tools/pmc_passes.sh <tag> [bench args]   -> gpurun_out/<tag>_hbm_pmc.csv, <tag>_hbm_traffic.json
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o r -- python3 bench.py --steps 4 --warmup 3 --eager --lean "$@" > gpurun_out/${TAG}_pmc_$C.log 2> gpurun_out/${TAG}_pmc_$C.err
  python3 tools/pmc_extract.py /tmp/pmc_$C/r_results.db $C gpurun_out/${TAG}_pmc_$C.csv 3 | tail -6
done
python3 tools/pmc_merge.py gpurun_out/${TAG}_pmc_FETCH_SIZE.csv gpurun_out/${TAG}_pmc_WRITE_SIZE.csv 3 gpurun_out/${TAG}_hbm_pmc.csv gpurun_out/${TAG}_hbm_traffic.json "${WORKLOAD:-S:192000}"
