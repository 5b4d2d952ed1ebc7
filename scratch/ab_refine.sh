# fused masked refinement (csrc/refine.hip): tests, then S-MM and XL-MM with / without
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_refine.py tests/test_gpu_model.py -x -q 2>&1 | tail -3 || exit 1
for v in fused tensor fused tensor; do
  if [ $v = tensor ]; then export AMC3D_NO_FUSED_REFINE=1; else unset AMC3D_NO_FUSED_REFINE; fi
  echo S-MM $v $(timeout -k 10 300 python bench.py --gpus 1 --mm --steps 32 --warmup 10 --lean 2>gpurun_out/refine_$v.err | tail -1 | cut -c1-60)
done
for v in fused tensor; do
  if [ $v = tensor ]; then export AMC3D_NO_FUSED_REFINE=1; else unset AMC3D_NO_FUSED_REFINE; fi
  echo XL-MM $v $(timeout -k 10 400 python bench.py --gpus 1 --steps 16 --warmup 8 --lean --variant XL --mm --batch 2 --points 64000 2>/dev/null | tail -1 | cut -c1-60)
done
