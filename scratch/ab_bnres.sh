# relu(bn(x) + identity) of the InvResMLP blocks inside the BatchNorm kernels: tests, then PointNeXt-L fused / unfused
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bn.py tests/test_gpu_model.py -x -q 2>&1 | tail -3 || exit 1
for v in fused unfused fused unfused; do
  if [ $v = unfused ]; then export AMC3D_NO_BN_RESIDUAL=1; else unset AMC3D_NO_BN_RESIDUAL; fi
  echo L $v $(timeout -k 10 300 python bench.py --gpus 1 --variant L --steps 24 --warmup 6 --lean 2>gpurun_out/bnres_$v.err | tail -1 | cut -c1-90)
done
unset AMC3D_NO_BN_RESIDUAL
