# cell-edge sweep of the thread-per-query 3-NN (AMC3D_NN3_CELL x the calibrated edge): parity once, then launch times
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "three_nn" 2>&1 | tail -2 || exit 1
for c in 0.7 1.0 1.4; do
  export AMC3D_NN3_CELL=$c
  echo cell $c
  bash scratch/prof_calls.sh nn3_grid | cut -c1-40
done
