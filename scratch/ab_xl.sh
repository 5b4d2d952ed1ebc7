# XL-MM side configurations: optimizer fold, library GEMMs vs this library's kernels on the deep short layers
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_optim.py -x -q 2>&1 | tail -2 || exit 1
for v in lib own lib own; do
  if [ $v = own ]; then export AMC3D_NO_LIBRARY_GEMM=1; else unset AMC3D_NO_LIBRARY_GEMM; fi
  echo XL-MM 2x64000 $v $(timeout -k 10 400 python bench.py --gpus 1 --steps 16 --warmup 8 --lean --variant XL --mm --batch 2 --points 64000 2>gpurun_out/xl_$v.err | tail -1 | cut -c1-90)
done
unset AMC3D_NO_LIBRARY_GEMM
