set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_syncbn.py tests/test_gpu_dist_bench.py -x -q > gpurun_out/sync_tests.log 2>&1 || { tail -40 gpurun_out/sync_tests.log; exit 1; }
tail -2 gpurun_out/sync_tests.log
for i in 1 2 3; do
echo pp$i $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --lean 2>gpurun_out/sync_pp$i.err | tail -1 | cut -c1-70)
done
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_v7_bench.json 2> gpurun_out/r2_v7_bench.err
tail -1 gpurun_out/r2_v7_bench.json | cut -c1-600
