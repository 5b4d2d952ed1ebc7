cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo XL-MM 1x120000 bf16 auto $(timeout -k 10 500 python bench.py --gpus 1 --steps 48 --warmup 24 --lean --variant XL --mm --batch 1 --points 120000 --dtype bf16 2>/dev/null | tail -1 | cut -c1-70)
done
echo XL-MM 1x120000 fp32 auto $(timeout -k 10 500 python bench.py --gpus 1 --steps 48 --warmup 24 --lean --variant XL --mm --batch 1 --points 120000 2>/dev/null | tail -1 | cut -c1-70)
echo XL-MM 2x64000 auto $(timeout -k 10 500 python bench.py --gpus 1 --steps 42 --warmup 21 --lean --variant XL --mm --batch 2 --points 64000 2>/dev/null | tail -1 | cut -c1-70)
