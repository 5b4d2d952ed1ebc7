cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_lagg.py tests/test_gpu_model.py tests/test_gpu_eval.py -x -q 2>&1 | tail -2 || exit 1
for v in a b; do
  echo S $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 48 --warmup 8 --lean 2>/dev/null | tail -1 | cut -c1-60)
done
for v in a b; do
  echo L $v $(timeout -k 10 300 python bench.py --gpus 1 --variant L --steps 24 --warmup 6 --lean 2>/dev/null | tail -1 | cut -c1-60)
done
