cd $GRAFT_REPO_ROOT
run() { echo "[$1] $(env $1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/main.err | tail -1 | cut -c1-60)"; }
run "A=1"
run "AMC3D_MAIN_CUS=0:240 AMC3D_CU_MASK=geo:0:144,fps:240:16"
run "AMC3D_MAIN_CUS=0:256 AMC3D_CU_MASK=geo:0:144"
run "AMC3D_MAIN_CUS=0:232 AMC3D_CU_MASK=geo:0:144,fps:232:24"
run "AMC3D_CU_MASK=geo:0:144,fps:240:16"
run "A=1"
