# scratch/res_iter.sh: unit tests of the fused residual branch, then its launches in one eager step
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_sa_res.py -x -q 2>&1 | tail -3 || exit 1
bash scratch/prof_calls.sh sa_res | cut -c1-60
