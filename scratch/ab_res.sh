# fused SetAbstraction residual branch (csrc/sa_res.hip) vs the torch operators: tests, then A/B of the step
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sa_res.py -x -q 2>&1 | tail -15 || exit 1
for v in fused torch fused torch; do
  if [ $v = torch ]; then export AMC3D_NO_SA_RESIDUAL=1; else unset AMC3D_NO_SA_RESIDUAL; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/res_$v.err | tail -1 | cut -c1-90)
done
unset AMC3D_NO_SA_RESIDUAL
