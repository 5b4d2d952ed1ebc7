set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for v in split nosplit split nosplit; do
  if [ $v = nosplit ]; then export AMC3D_NO_SPLIT_K=1; else unset AMC3D_NO_SPLIT_K; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/sk_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_NO_SPLIT_K
bash scratch/prof_calls.sh gm_gemm gm_split
