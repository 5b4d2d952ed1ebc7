cd $GRAFT_REPO_ROOT
for v in lib own lib own; do
  if [ $v = own ]; then export AMC3D_WGRAD_OWN=1; else unset AMC3D_WGRAD_OWN; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/wgrad_$v.err | tail -1 | cut -c1-90)
done
