cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do
  AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 12 --warmup 6 --lean > gpurun_out/sr_$i.out 2> gpurun_out/sr_$i.err; echo "run $i rc=$? $(tail -1 gpurun_out/sr_$i.out | cut -c1-80)"
done
timeout -k 10 900 python -m pytest tests/test_gpu_syncbn.py tests/test_gpu_dist_bench.py -x -q 2>&1 | tail -2
