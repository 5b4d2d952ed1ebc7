set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py -x -q 2>&1 | tail -3
for v in on off on off; do
  if [ $v = off ]; then export AMC3D_NO_STREAMING_WGRAD=1; else unset AMC3D_NO_STREAMING_WGRAD; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/gw_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_NO_STREAMING_WGRAD
bash scratch/prof_calls.sh gw_wgrad "gm_gemm_kernel<true, true"
