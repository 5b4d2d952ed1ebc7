cd $GRAFT_REPO_ROOT
for v in all std all std; do
  if [ $v = all ]; then export AMC3D_FPS_ALL=1; else unset AMC3D_FPS_ALL; fi
  echo S-$v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/all_$v.err | tail -1 | cut -c1-60)
done
export AMC3D_FPS_ALL=1
echo L-all $(timeout -k 10 300 python bench.py --gpus 1 --steps 24 --warmup 8 --lean --variant L 2>gpurun_out/all_L.err | tail -1 | cut -c1-60)
echo S-all-mask128 $(AMC3D_CU_MASK=geo:0:128 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/all_m.err | tail -1 | cut -c1-60)
echo S-all-mask160 $(AMC3D_CU_MASK=geo:0:160 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/all_m.err | tail -1 | cut -c1-60)
