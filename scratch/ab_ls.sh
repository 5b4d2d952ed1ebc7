set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_lagg.py -x -q 2>&1 | tail -3
for v in one four one four; do
  if [ $v = four ]; then export AMC3D_LAGG_SCATTER_TILES4=1; else unset AMC3D_LAGG_SCATTER_TILES4; fi
  echo L-$v $(timeout -k 10 300 python bench.py --gpus 1 --steps 24 --warmup 8 --lean --variant L 2>gpurun_out/ls_$v.err | tail -1 | cut -c1-70)
done
