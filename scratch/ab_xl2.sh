cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_gcc.py -x -q 2>&1 | tail -2 || exit 1
for v in a b; do
  echo XL-MM 2x64000 $v $(timeout -k 10 400 python bench.py --gpus 1 --steps 16 --warmup 8 --lean --variant XL --mm --batch 2 --points 64000 2>gpurun_out/xl_$v.err | tail -1 | cut -c1-90)
done
echo XL-MM 1x120000 bf16 $(timeout -k 10 400 python bench.py --gpus 1 --steps 16 --warmup 8 --lean --variant XL --mm --batch 1 --points 120000 --dtype bf16 2>gpurun_out/xl_c.err | tail -1 | cut -c1-90)
