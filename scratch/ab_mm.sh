set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py -x -q 2>&1 | tail -2
for i in 1 2; do echo MM $(timeout -k 10 300 python bench.py --gpus 1 --steps 24 --warmup 8 --lean --mm 2>gpurun_out/mm.err | tail -1 | cut -c1-70); done
echo S $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/mm.err | tail -1 | cut -c1-70)
bash scratch/prof_calls_all.sh 30 --mm | grep "pw_wgrad\|reduce_kernel\|vectorized_gather" | cut -c1-140
