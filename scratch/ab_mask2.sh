# re-sweep of the geometry queue's CU mask (four-lane schedule)
cd $GRAFT_REPO_ROOT
for n in 128 144 160 176 128 144 160 176; do
  export AMC3D_CU_MASK="geo:0:$n"
  echo geo $n $(timeout -k 10 300 python bench.py --gpus 1 --steps 48 --warmup 8 --lean 2>gpurun_out/mask_$n.err | tail -1 | cut -c1-60)
done
