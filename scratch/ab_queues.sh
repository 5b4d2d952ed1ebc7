set -e
cd $GRAFT_REPO_ROOT

run() { echo "$1 | $2 | $(timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean $2 2>gpurun_out/q.err | tail -1 | cut -c1-60)"; }
unset AMC3D_QUEUES; run default ""
export AMC3D_QUEUES=fps0,fps1,geo; run "$AMC3D_QUEUES" ""
export AMC3D_QUEUES=fps0,fps1,a2,b; run "$AMC3D_QUEUES" ""
export AMC3D_QUEUES=fps0,fps1,fps2,geo; run "$AMC3D_QUEUES" "--fps-lanes 3"
unset AMC3D_QUEUES; run default ""
unset AMC3D_NO_CONTRAST_CSR
export AMC3D_QUEUES=fps0,fps1,geo; run "csr $AMC3D_QUEUES" ""
