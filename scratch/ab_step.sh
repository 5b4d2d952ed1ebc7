# scratch/ab_step.sh [pytest args...]: tests given on the command line, two lean bench runs
cd $GRAFT_REPO_ROOT
if [ $# -gt 0 ]; then timeout -k 10 900 python -m pytest "$@" -x -q 2>&1 | tail -3 || exit 1; fi
for v in a b; do
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/step_$v.err | tail -1 | cut -c1-90)
done
