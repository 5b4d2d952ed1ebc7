cd $GRAFT_REPO_ROOT
for v in first side first side; do
  if [ $v = side ]; then export AMC3D_SIDE_FIRST=1; else unset AMC3D_SIDE_FIRST; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/first_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_SIDE_FIRST
echo sync $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>gpurun_out/first_sync.err | tail -1 | cut -c1-70)
echo sync-sidefirst $(AMC3D_SIDE_FIRST=1 AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2>gpurun_out/first_sync2.err | tail -1 | cut -c1-70)
AMC3D_TIMELINE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2> gpurun_out/timeline7.err; grep timeline gpurun_out/timeline7.err
