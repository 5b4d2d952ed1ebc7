cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bn.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|Error" | head -5
for v in fused plain fused plain; do
  if [ $v = plain ]; then export AMC3D_NO_BN_SIGMOID=1; else unset AMC3D_NO_BN_SIGMOID; fi
  echo S-MM $v $(timeout -k 10 300 python bench.py --gpus 1 --mm --steps 32 --warmup 10 --lean 2>gpurun_out/bnsig_$v.err | tail -1 | cut -c1-60)
done
