cd $GRAFT_REPO_ROOT
run() { echo "[$1 | $2] $(env $1 timeout -k 10 400 python bench.py --gpus 1 --steps 16 --warmup 8 --lean $2 2>gpurun_out/xl.err | tail -1 | cut -c1-60)"; }
XL="--variant XL --mm --batch 2 --points 64000"
XL5="--variant XL --mm --batch 1 --points 120000 --dtype bf16"
run "A=1" "$XL5"
run "A=1" "$XL"
run "A=1" ""
run "A=1" "--variant L"
