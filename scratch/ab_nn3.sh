# thread-per-query 3-NN: parity, the step, the launches
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "three_nn or interpol" 2>&1 | tail -4 || exit 1
for v in a b; do
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/nn3_$v.err | tail -1 | cut -c1-90)
done
bash scratch/prof_calls.sh nn3_grid | cut -c1-60
