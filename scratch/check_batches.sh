cd $GRAFT_REPO_ROOT
export AMC3D_CHECK_BATCHES=1
for l in 2 4 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 2 --lean --fps-lanes $l 2>&1 >/dev/null | grep -E "batch check|Error|error|assert" | head -3
done
timeout -k 10 600 python bench.py --gpus 1 --steps 12 --warmup 2 --lean --variant XL --mm --batch 1 --points 120000 2>&1 >/dev/null | grep -E "batch check|Error|error|assert" | head -3
timeout -k 10 600 python bench.py --gpus 1 --steps 7 --warmup 2 --lean --variant XL --mm --batch 2 --points 64000 2>&1 >/dev/null | grep -E "batch check|Error|error|assert" | head -3
