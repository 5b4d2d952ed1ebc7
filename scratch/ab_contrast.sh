set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_loss.py -x -q > gpurun_out/c1_loss.log 2>&1 || { tail -30 gpurun_out/c1_loss.log; exit 1; }
tail -2 gpurun_out/c1_loss.log
for v in csr nocsr csr nocsr; do
  if [ $v = nocsr ]; then export AMC3D_NO_CONTRAST_CSR=1; else unset AMC3D_NO_CONTRAST_CSR; fi
  timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean > gpurun_out/c1_bench_$v.log 2>gpurun_out/c1_bench.err && echo $v $(tail -1 gpurun_out/c1_bench_$v.log | cut -c1-100)
done
