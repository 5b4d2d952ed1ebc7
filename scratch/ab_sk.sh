set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
bash scratch/ab_own.sh
