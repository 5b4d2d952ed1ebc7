cd $GRAFT_REPO_ROOT
run() { echo "a2fps=[$1] mask=[$2] $(AMC3D_A2_ON_FPS=$1 AMC3D_CU_MASK=$2 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/a2.err | tail -1 | cut -c1-40)"; }
run "1" "geo:0:128"
run "1" "geo:0:112"
run "1" "geo:0:96"
run "1" "geo:0:80"
run "1" "geo:0:144"
run "1" "geo:64:128"
run "1" "geo:0:128,fps:128:128"
run "1" "geo:0:128"
