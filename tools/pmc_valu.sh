#!/bin/bash
# Which kernels of the train step are bound by vector-ALU issue rather than by memory: one rocprofv3 --pmc pass of SQ counters over
# the eager step (every dispatch one kernel), summarised per kernel (tools/pmc_valu_merge.py):
#   valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)   (quad-cycles of VALU execution per SIMD cycle)
#   wait      = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                    (share of wave time parked on s_waitcnt / barriers)
# usage: tools/pmc_valu.sh <tag> [bench args]  -> gpurun_out/<tag>_valu_pmc.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_valu
timeout -k 10 500 rocprofv3 --pmc $CS --kernel-trace -d /tmp/pmc_valu -o r -- python3 bench.py --steps 4 --warmup 3 --eager --lean "$@" > gpurun_out/${TAG}_pmc_valu.log 2> gpurun_out/${TAG}_pmc_valu.err
for C in $CS; do python3 tools/pmc_extract.py /tmp/pmc_valu/r_results.db $C gpurun_out/${TAG}_pmc_$C.csv 3 > /dev/null; done
python3 tools/pmc_valu_merge.py gpurun_out/${TAG} 3 gpurun_out/${TAG}_valu_pmc.csv
