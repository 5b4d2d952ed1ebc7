set -e
cd $GRAFT_REPO_ROOT
for v in pp nopp pp nopp; do
  if [ $v = nopp ]; then export AMC3D_NO_PINGPONG=1; else unset AMC3D_NO_PINGPONG; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/pp_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_NO_PINGPONG
echo syncbn $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --lean 2>gpurun_out/pp_sync.err | tail -1 | cut -c1-70)
echo L $(timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --lean --variant L 2>gpurun_out/pp_L.err | tail -1 | cut -c1-70)
