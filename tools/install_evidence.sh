#!/bin/bash
# tools/install_evidence.sh <tag> <version>: copy what tools/round_evidence.sh <tag> left in gpurun_out/ into profiles/ -- under
# round3_<version>_* (kept per run) and under the canonical names amcontrast3d_amd/roofline.py reads
set -e
T=$1; V=$2; G=gpurun_out; P=profiles
cp $G/${T}S_kernel_stats.csv    $P/round3_${V}_S_kernel_stats.csv;    cp $G/${T}S_kernel_stats.csv    $P/round3_kernel_stats.csv
cp $G/${T}L_kernel_stats.csv    $P/round3_${V}_L_kernel_stats.csv;    cp $G/${T}L_kernel_stats.csv    $P/round3_L_kernel_stats.csv
cp $G/${T}XLMM_kernel_stats.csv $P/round3_${V}_XLMM_kernel_stats.csv; cp $G/${T}XLMM_kernel_stats.csv $P/round3_XL_kernel_stats.csv
cp $G/${T}S_steady.txt $P/round3_${V}_S_steady_state.txt; cp $G/${T}L_steady.txt $P/round3_${V}_L_steady_state.txt
cp $G/${T}XLMM_steady.txt $P/round3_${V}_XLMM_steady_state.txt
cp $G/${T}S_hbm_pmc.csv $P/round3_${V}_hbm_pmc.csv; cp $G/${T}S_hbm_pmc.csv $P/round3_hbm_pmc.csv
cp $G/${T}S_hbm_traffic.json $P/hbm_traffic.json
cp $G/${T}S_valu_pmc.csv $P/round3_${V}_valu_pmc.csv; cp $G/${T}S_valu_pmc.csv $P/round3_valu_pmc.csv
tail -1 $G/${T}_bench.json > $P/round3_${V}_bench.json
cp $G/${T}_side_configs.jsonl $P/round3_${V}_side_configs.jsonl
ls -la $P | grep "round3_${V}\|hbm_traffic\|round3_kernel\|round3_valu"
