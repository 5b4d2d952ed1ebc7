cd $GRAFT_REPO_ROOT
run() { echo "mask=[$1] $(AMC3D_CU_MASK=$1 timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/mask.err | tail -1 | cut -c1-40)"; }
run "geo:0:192"
run "geo:0:160"
run "geo:0:176"
run "geo:0:208"
run "geo:0:224"
run "geo:32:192"
run "geo:0:192,fps:192:64"
run "geo:0:192,fps:0:192"
run "fps:192:64"
run ""
