cd $GRAFT_REPO_ROOT
run() { echo "[$1 | $2] $(env $1 timeout -k 10 400 python bench.py --gpus 1 --steps 12 --warmup 6 --lean $2 2>gpurun_out/xl.err | tail -1 | cut -c1-60)"; }
XL="--variant XL --mm --batch 2 --points 64000"
run "A=1" "$XL"
run "A=1" "$XL --fps-lanes 2"
run "A=1" "$XL --fps-lanes 3"
run "A=1" "$XL --fps-lanes 4"
run "AMC3D_CU_MASK=geo:0:144" "$XL --fps-lanes 3"
run "AMC3D_NO_FPS_JOINT=1" "$XL"
run "A=1" ""
XL5="--variant XL --mm --batch 1 --points 120000 --dtype bf16"
run "A=1" "$XL5"
run "A=1" "$XL5 --fps-lanes 4"
