cd $GRAFT_REPO_ROOT
run() { echo "$1 | $(env $2 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean $3 2>gpurun_out/own.err | tail -1 | cut -c1-60)"; }
run "S default" "A=1" ""
run "S own-all" "AMC3D_NO_LIBRARY_GEMM=1 AMC3D_SMALL_CONV_OWN=1" ""
run "S own-small" "AMC3D_SMALL_CONV_OWN=1" ""
run "S default" "A=1" ""
run "L default" "A=1" "--variant L"
run "L own-all" "AMC3D_NO_LIBRARY_GEMM=1 AMC3D_SMALL_CONV_OWN=1" "--variant L"
