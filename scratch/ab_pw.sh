set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
for i in 1 2; do echo S $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>/dev/null | tail -1 | cut -c1-60); done
bash scratch/prof_calls.sh "pw_gemm_kernel<2>" "pw_gemm_kernel<4>" | cut -c1-140
