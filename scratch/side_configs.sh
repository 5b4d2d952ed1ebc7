#!/bin/bash
# side configurations (lean lines only) -> gpurun_out/<tag>_side_configs.jsonl   usage: scratch/side_configs.sh <tag>
TAG=$1
cd $GRAFT_REPO_ROOT
( echo -n '{"config": "S-MM 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --mm 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "L 8x24000", "line": '; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --variant L 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 2x64000 (cfg 4 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 42 --warmup 21 --lean --variant XL --mm --batch 2 --points 64000 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "XL-MM 1x120000 bf16 (cfg 5 per GPU)", "line": '; timeout -k 10 400 python bench.py --gpus 1 --steps 48 --warmup 24 --lean --variant XL --mm --batch 1 --points 120000 --dtype bf16 2>/dev/null | tail -1; echo '}'
  echo -n '{"config": "S 8x24000, SyncBatchNorm segmentation with a one-rank RCCL group", "line": '; AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>/dev/null | tail -1; echo '}'
) > gpurun_out/${TAG}_side_configs.jsonl
cat gpurun_out/${TAG}_side_configs.jsonl
