cd $GRAFT_REPO_ROOT
run() { echo "[$1] $(timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 4 --lean $1 2>gpurun_out/var_check.err | tail -1 | cut -c1-70)  rc=$?"; }
run "--pool 1"
run "--pool 3"
run "--pool 5"
run "--no-overlap"
run "--fps-lanes 1"
run "--fps-lanes 3"
run "--dtype bf16"
run "--no-graph"
run "--mm"
run "--variant B"
run "--batch 2 --points 4096"
echo "[eval] $(timeout -k 10 300 python bench.py --eval 2>gpurun_out/var_eval.err | tail -1 | cut -c1-100)"
