cd $GRAFT_REPO_ROOT
for v in 8 4 2 8 4; do
  echo minc=$v $(AMC3D_GW_MINCHUNKS=$v timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/minc.err | tail -1 | cut -c1-60)
done
AMC3D_GW_MINCHUNKS=4 bash scratch/prof_calls.sh gw_wgrad | cut -c1-150
