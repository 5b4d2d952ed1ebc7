cd $GRAFT_REPO_ROOT
echo new $(AMC3D_CHECK_VARIANTS=1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/in.err | tail -1 | cut -c1-70); grep "gradient norms" gpurun_out/in.err
echo new $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/in2.err | tail -1 | cut -c1-70)
echo nopp $(AMC3D_NO_PINGPONG=1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/in3.err | tail -1 | cut -c1-70)
echo sync $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>gpurun_out/in4.err | tail -1 | cut -c1-70)
timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 5 > gpurun_out/in_full.json 2> gpurun_out/in_full.err; tail -1 gpurun_out/in_full.json | cut -c1-120
