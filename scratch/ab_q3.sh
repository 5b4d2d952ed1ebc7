cd $GRAFT_REPO_ROOT
run() { echo "queues=[$1] mask=[$2] $(AMC3D_QUEUES=$1 AMC3D_CU_MASK=$2 timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/q3.err | tail -1 | cut -c1-40)"; }
run "fps,geo" "geo:0:192"
run "fps,a2,b" "b:0:192"
run "fps,a2,b" "b:0:128"
run "fps,a2,b" "b:0:96"
run "fps,a2,b" "b:0:160"
run "fps,a2,b" "b:0:128,a2:128:32,fps:160:32"
