cd $GRAFT_REPO_ROOT
for v in default own default own; do
  if [ $v = own ]; then export AMC3D_WGRAD_FORM=own; else unset AMC3D_WGRAD_FORM; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 48 --warmup 8 --lean 2>gpurun_out/wgrad_$v.err | tail -1 | cut -c1-60)
done
