set -e
cd $GRAFT_REPO_ROOT
echo S $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/chk_S.err | tail -1 | cut -c1-70)
echo L $(timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --variant L 2>gpurun_out/chk_L.err | tail -1 | cut -c1-70)
echo sync $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>gpurun_out/chk_sync.err | tail -1 | cut -c1-70)
echo MM $(timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean --mm 2>gpurun_out/chk_mm.err | tail -1 | cut -c1-70)
timeout -k 10 900 python -m pytest tests/test_gpu_dist_bench.py -x -q 2>&1 | tail -2
