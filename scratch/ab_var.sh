cd $GRAFT_REPO_ROOT
echo fixed $(AMC3D_CHECK_VARIANTS=1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/var.err | tail -1 | cut -c1-70); grep "gradient norms" gpurun_out/var.err
echo nopp $(AMC3D_NO_PINGPONG=1 AMC3D_CHECK_VARIANTS=1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/var2.err | tail -1 | cut -c1-70); grep "gradient norms" gpurun_out/var2.err
echo sync $(AMC3D_FORCE_SYNC_BN=1 AMC3D_CHECK_VARIANTS=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>gpurun_out/var3.err | tail -1 | cut -c1-70); grep "gradient norms" gpurun_out/var3.err
