cd $GRAFT_REPO_ROOT
for v in fused torch fused torch; do
  if [ $v = torch ]; then export AMC3D_TORCH_ADAMW=1; else unset AMC3D_TORCH_ADAMW; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/opt_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_TORCH_ADAMW
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dist_bench.py tests/test_gpu_optim.py -x -q 2>&1 | tail -3
