set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sa_tail.py tests/test_gpu_lagg.py tests/test_gpu_loss.py -x -q > gpurun_out/pm_tests.log 2>&1 || { tail -40 gpurun_out/pm_tests.log; exit 1; }
tail -2 gpurun_out/pm_tests.log

for v in pm cm pm cm; do
  if [ $v = cm ]; then export AMC3D_SAT_DX1_CM=1; else unset AMC3D_SAT_DX1_CM; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/pm.err | tail -1 | cut -c1-60)
done
