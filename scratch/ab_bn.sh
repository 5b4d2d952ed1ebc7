# single-launch BatchNorm for the short deep layers: tests, then the step
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bn.py tests/test_gpu_model.py tests/test_gpu_graph.py -x -q 2>&1 | tail -4 || exit 1
for v in a b; do
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/bn_$v.err | tail -1 | cut -c1-90)
done
