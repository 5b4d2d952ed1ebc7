cd $GRAFT_REPO_ROOT
for v in joint nojoint joint nojoint; do
  if [ $v = nojoint ]; then export AMC3D_NO_FPS_JOINT=1; else unset AMC3D_NO_FPS_JOINT; fi
  echo $v $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/joint_$v.err | tail -1 | cut -c1-70)
done
unset AMC3D_NO_FPS_JOINT
AMC3D_TIMELINE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2> gpurun_out/timeline4.err; grep timeline gpurun_out/timeline4.err
