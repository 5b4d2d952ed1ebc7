set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for v in pipe nopipe pipe nopipe; do
  if [ $v = nopipe ]; then export AMC3D_NO_PW_PIPE=1; else unset AMC3D_NO_PW_PIPE; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/gw_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_NO_PW_PIPE
bash scratch/prof_calls.sh pw_gemm
