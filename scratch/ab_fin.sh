set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sa_tail.py tests/test_gpu_pwconv.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for i in 1 2; do echo new $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/fin.err | tail -1 | cut -c1-70); done
bash scratch/prof_calls.sh sat_finalize gcc_reduce_partials | cut -c1-150
