cd $GRAFT_REPO_ROOT
run() { echo "queues=[$1] mask=[$2] $(AMC3D_QUEUES=$1 AMC3D_CU_MASK=$2 timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/q2.err | tail -1 | cut -c1-40)"; }
run "fps,geo" "geo:0:192"
run "fps0,fps1,geo" "geo:0:192"
run "fps0,fps1,geo" "geo:0:192,fps0:192:32,fps1:224:32"
run "fps,geo" "geo:0:192"
AMC3D_TIMELINE=1 AMC3D_QUEUES=fps0,fps1,geo timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2> gpurun_out/timeline2.err; grep timeline gpurun_out/timeline2.err
