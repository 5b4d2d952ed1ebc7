#!/bin/bash
# the round's evidence in one GPU call: tools/round_evidence.sh <tag>
#   steady-state kernel tables (S, L, XL-MM 2x64000), the HBM counter passes (S), the driver's bench command, side configurations
set -e
TAG=$1
cd $GRAFT_REPO_ROOT
LINES_SHOWN=3 bash tools/prof_steady.sh ${TAG}S
LINES_SHOWN=3 bash tools/prof_steady.sh ${TAG}L --variant L
LINES_SHOWN=3 bash tools/prof_steady.sh ${TAG}XLMM --variant XL --mm --batch 2 --points 64000 --steps 24 --warmup 24
bash tools/pmc_passes.sh ${TAG}S | tail -4
bash tools/pmc_valu.sh ${TAG}S | head -12
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-200
( echo -n '{"config": "S-MM 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --mm 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "L 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --variant L 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 2x64000 (cfg 4 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 42 --warmup 28 --lean --variant XL --mm --batch 2 --points 64000 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 1x120000 bf16 (cfg 5 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 48 --warmup 30 --lean --variant XL --mm --batch 1 --points 120000 --dtype bf16 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 1x120000 fp32", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 48 --warmup 30 --lean --variant XL --mm --batch 1 --points 120000 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "S 8x24000, SyncBatchNorm segmentation with a one-rank RCCL group", "line": '; AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>/dev/null | tail -1; echo '}'
) > gpurun_out/${TAG}_side_configs.jsonl
cat gpurun_out/${TAG}_side_configs.jsonl
