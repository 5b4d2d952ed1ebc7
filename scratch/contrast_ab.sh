set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -3
echo "--- product"; timeout -k 10 200 python scratch/contrast_bench.py 2>/dev/null | grep stage
echo "--- no neighbour-row atomics"; AMC3D_LIB=$GRAFT_REPO_ROOT/scratch/diag/libamc3d_contrast1.so timeout -k 10 200 python scratch/contrast_bench.py 2>/dev/null | grep stage
