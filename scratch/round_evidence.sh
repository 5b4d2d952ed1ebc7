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
# side configurations (lean lines only)
( echo -n '{"config": "S-MM 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --mm 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "L 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --variant L 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 2x64000 (cfg 4 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 42 --warmup 21 --lean --variant XL --mm --batch 2 --points 64000 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 1x120000 bf16 (cfg 5 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 48 --warmup 24 --lean --variant XL --mm --batch 1 --points 120000 --dtype bf16 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "S 8x24000, SyncBatchNorm segmentation with a one-rank RCCL group", "line": '; AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>/dev/null | tail -1; echo '}'
) > gpurun_out/${TAG}_side_configs.jsonl
cat gpurun_out/${TAG}_side_configs.jsonl
