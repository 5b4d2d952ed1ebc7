# NOTE: the switch this script toggles (an experiment of round 2) was measured and removed again; kept for the record of how it was measured
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_lagg.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for v in on off on off; do
  if [ $v = off ]; then export AMC3D_NO_DP_EDGES=1; else unset AMC3D_NO_DP_EDGES; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/dpe_$v.err | tail -1 | cut -c1-70)
done
