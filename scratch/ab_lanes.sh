cd $GRAFT_REPO_ROOT
export AMC3D_CU_MASK="geo:0:160"
for cfg in "--variant L" "--mm"; do
  for l in 2 4 2 4; do
    echo "$cfg" lanes $l $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 12 --lean --fps-lanes $l $cfg 2>gpurun_out/lanes_$l.err | tail -1 | cut -c1-60)
  done
done
